"""The DDPM and SD data-parallel paths on the REAL kernels: two ranks sharing the test box's GPU (backend gloo through
SALUN_DIST_BACKEND, since RCCL refuses two ranks on one device) against one rank on the same global batches — the same
worker scripts as tests/test_dist_diffusion_gloo.py, here with the MFMA convolutions, the ResnetBlock nodes,
`salun_dropout` (keep decisions keyed by the global sample index: the two ranks' masks ARE the single process's rows),
the fused Adam and the bucketed gradient reducer.  Ranks must stay bit-identical and equal the single-process run to
fp32 summation tolerance (reference being replaced: nn.DataParallel, DDPM/runners/diffusion.py:504,582-593,948-996)."""
import os

import numpy as np
import pytest

import test_dist_diffusion_gloo as G

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _on_device(monkeypatch):
    monkeypatch.setenv("SALUN_TEST_DEVICE", "cuda")


def test_ddpm_two_ranks_on_one_gpu_equal_one_rank(tmp_path):
    G.test_ddpm_two_ranks_equal_one_rank_with_dropout_and_label_drop(tmp_path)


def test_sd_two_ranks_on_one_gpu_equal_one_rank(tmp_path):
    G.test_sd_two_ranks_equal_one_rank_over_sharded_global_batches(tmp_path)


def test_draws_on_the_device_follow_the_global_sample_index(tmp_path):
    G.test_two_ranks_see_the_single_process_draws_of_their_own_samples(tmp_path)
