"""Host side of the Classification drivers (main_random / main_forget / generate_mask) up to the point where a
device is needed: argument parsing, model + synthetic dataset construction, the reference's 10 % random-forget
marking, forget / retain split, loaders, test-transform switch.  CPU."""
import numpy as np
import torch


def test_synthetic_pipeline_builds_the_reference_shaped_workload():
    from unlearn_saliency_amd.Classification import _driver, arg_parser, utils
    from unlearn_saliency_amd.Classification.dataset import BatchLoader, split_marked
    args = arg_parser.parse_args(["--synthetic", "--unlearn", "RL", "--save_dir", "/tmp/unused", "--mask_path", "x",
                                  "--num_indexes_to_replace", "4500"])
    utils.setup_seed(args.seed)
    model, train_full, val_loader, test_loader, marked = utils.setup_model_dataset(args)
    assert sum(p.numel() for p in model.parameters()) == 11_173_962
    assert len(list(model.named_parameters())) == 62 and len(model.state_dict()) >= 122
    assert (len(train_full.dataset), len(val_loader.dataset), len(test_loader.dataset)) == (45000, 5000, 10000)
    forget, retain = split_marked(marked.dataset)
    assert (len(forget), len(retain)) == (4500, 40500)
    # the forget indices are the reference's draw: RandomState(seed - 1).choice(45000, 4500, replace=False)
    want = np.random.RandomState(args.seed - 1).choice(45000, 4500, replace=False)
    assert np.array_equal(np.sort(want), np.sort(np.where(np.asarray(marked.dataset.targets) < 0)[0]))
    x, y = next(iter(BatchLoader(forget, 256, True)))
    assert x.shape == (256, 3, 32, 32) and x.dtype == torch.float32 and 0.0 <= float(x.min()) and float(x.max()) <= 1.0
    assert y.dtype == torch.int64 and int(y.min()) >= 0 and int(y.max()) <= 9  # labels restored by the split
    utils.dataset_convert_to_test(forget, args)
    assert forget.transform == "test"
    assert len(_driver._head(retain, len(test_loader.dataset))) == 10000


def test_ddpm_config_and_class_forget_loaders():
    """YAML schema of configs/cifar10_saliency_unlearn.yml and the class-forget split (label 0: 5,000 of 50,000
    synthetic samples at batch 128 -> 40 forget / 352 remain batches)."""
    import os
    from types import SimpleNamespace
    from unlearn_saliency_amd.DDPM.datasets import get_forget_dataset
    from unlearn_saliency_amd.DDPM.functions import load_config
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = load_config(os.path.join(root, "unlearn_saliency_amd", "DDPM", "configs", "cifar10_saliency_unlearn.yml"))
    assert (cfg.model.ch, list(cfg.model.ch_mult), cfg.model.num_res_blocks) == (128, [1, 2, 2, 2], 2)
    assert (cfg.training.batch_size, cfg.diffusion.num_diffusion_timesteps, cfg.optim.lr) == (128, 1000, 1e-4)
    remain, forget = get_forget_dataset(SimpleNamespace(label_to_forget=0), cfg, 0, device=torch.device("cpu"),
                                        synthetic=True)
    assert (len(remain), len(forget)) == (352, 40)
    x, c = next(iter(forget))
    assert x.shape == (128, 3, 32, 32) and bool((c == 0).all())
    x, c = next(iter(remain))
    assert bool((c != 0).all())


def test_sd_unet_full_configuration_has_the_reference_parameter_table():
    """SD-v1 U-Net on the meta device: 686 tensors / 859,520,964 parameters (mask keys of SD/train-scripts)."""
    from unlearn_saliency_amd.SD.unet import V1_UNET_CONFIG, UNetModel
    with torch.device("meta"):
        m = UNetModel(**V1_UNET_CONFIG)
    names = [n for n, _ in m.named_parameters()]
    assert len(names) == 686 and sum(p.numel() for p in m.parameters()) == 859_520_964
    assert names[0] == "time_embed.0.weight" and any("attn2.to_k.weight" in n for n in names)


def test_unlearn_checkpoint_round_trip_and_file_names(tmp_path):
    """{save_dir}/{unlearn}checkpoint.pth.tar + {unlearn}eval_result.pth.tar, as the reference writes them
    (unlearn/impl.py:21-51, utils.py save_checkpoint)."""
    import os
    from types import SimpleNamespace
    from fixtures import TinyCNN, tiny_state
    from unlearn_saliency_amd.Classification import unlearn, utils
    m = TinyCNN()
    m.load_state_dict(tiny_state(3))
    args = SimpleNamespace(save_dir=str(tmp_path), unlearn="RL")
    unlearn.save_unlearn_checkpoint(m, {"accuracy": {"forget": 1.0}}, args)
    assert sorted(os.listdir(tmp_path)) == ["RLcheckpoint.pth.tar", "RLeval_result.pth.tar"]
    m2 = TinyCNN()
    got = unlearn.load_unlearn_checkpoint(m2, torch.device("cpu"), args)
    assert got is not None and got[1] == {"accuracy": {"forget": 1.0}}
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    assert unlearn.load_unlearn_checkpoint(m2, torch.device("cpu"), SimpleNamespace(save_dir=str(tmp_path), unlearn="GA")) is None
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    utils.warmup_lr(0, 5, opt, one_epoch_step=10, args=SimpleNamespace(warmup=2, lr=0.1, unlearn_lr=0.1))
    assert abs(opt.param_groups[0]["lr"] - 0.025) < 1e-12  # lr * step / (warmup * steps_per_epoch), reference utils.py:33-41
    assert float(utils.accuracy(torch.tensor([[0.1, 0.9], [0.8, 0.2]]), torch.tensor([1, 1]))[0]) == 50.0
