"""GPU, slow: the per-rank FULL sizes of BASELINE configs[3] and configs[4] pushed through the frame loops once
(tests/fullsize_cases.py): 12 500 frames of 1024^2 through the chunked pixel-series loop of the N > 1 path, and the 6 250 frame
sets of 4 cameras on the 5 M-triangle model that are a rank's share of configs[4] (50 000 sets on 8 GPUs) -- tens of GB resident, checked through size-independent properties."""
import pytest

pytestmark = pytest.mark.gpu


def _enough_memory(gb):
    import gc
    import torch
    gc.collect()
    torch.cuda.empty_cache()              # (what earlier tests of this process left in torch's cache is not "in use")
    free, _ = torch.cuda.mem_get_info()
    return free > gb * 1e9


def test_config3_rank_share_12500_frames(gpu_lib, oracle):
    import fullsize_cases
    if not _enough_memory(90):
        pytest.skip("needs 90 GB of free HBM")
    facts = fullsize_cases.config3_rank_share(oracle, verbose=True)
    assert facts["frames"] == 12500 and facts["exchange_chunks"] == 13
    assert facts["series_pitch_floats"] % 64 == 0          # rows on 256-byte boundaries


def test_config4_rank_share_6250_frame_sets(gpu_lib):
    import fullsize_cases
    if not _enough_memory(170):
        pytest.skip("needs 170 GB of free HBM")
    facts = fullsize_cases.config4_rank_share(verbose=True)
    assert facts["frame_sets"] == 6250 and facts["triangles"] > 4_900_000
