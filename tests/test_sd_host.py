"""Host-only checks of the SD front end (no kernel runs)."""
import torch
import torch.nn as nn

from fixtures import sd_tiny_config


def test_frozen_param_count_comes_from_the_model_not_from_setup_model():
    """the proximal step ranks over U-Net + frozen stages (reference proximal_gradient.py:66-72,141-167);
    the count must not depend on which helper built the model."""
    from unlearn_saliency_amd.SD.ldm_lite import SD_V1_FROZEN_PARAMS, LatentDiffusionLite
    assert SD_V1_FROZEN_PARAMS == 83_653_863 + 123_060_480
    with torch.device("meta"):
        v1 = LatentDiffusionLite()  # the v1-inference shape, no stages attached: the SD-v1 constant
        assert v1.frozen_param_count == SD_V1_FROZEN_PARAMS
        tiny = LatentDiffusionLite(sd_tiny_config())
        assert tiny.frozen_param_count == 0
        stage = nn.Linear(10, 7)  # attached stages are counted for what they are
        assert LatentDiffusionLite(sd_tiny_config(), first_stage=stage, cond_stage=lambda t: t).frozen_param_count == 77
        assert LatentDiffusionLite(first_stage=stage, cond_stage=nn.Linear(3, 3)).frozen_param_count == 77 + 12


def test_sharded_batches_partition_every_global_batch():
    from unlearn_saliency_amd.SD.train_scripts import ShardedBatches
    batches = [(torch.arange(5 * 2).view(5, 2), torch.arange(5)), (torch.arange(4 * 2).view(4, 2), torch.arange(4))]
    parts = [list(ShardedBatches(batches, r, 3)) for r in range(3)]
    for i, full in enumerate(batches):
        assert torch.equal(torch.cat([parts[r][i][0] for r in range(3)]), full[0])
        assert torch.equal(torch.cat([parts[r][i][1] for r in range(3)]), full[1])
    sb = ShardedBatches(batches, 1, 3)
    it = iter(sb)
    next(it)
    assert sb.last_shard == (1, 3, 5) and len(sb) == 2
