"""BASELINE.json configs[4] (SURVEY.md section 8d "Stream"): chunks produced on the device, sharded by contiguous
chunk, one reduction of the metric partials.  On one GPU the ranks are emulated one after the other."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def runner():
    from umetrack_amd.stream import StreamRunner
    return StreamRunner(torch.device("cuda", 0), chunk_frames=128, frames_per_recording=64)


def test_chunk_is_a_function_of_its_index(runner):
    a = runner.run_chunk(3)
    b = runner.run_chunk(1)
    c = runner.run_chunk(3)
    assert torch.equal(a, c) and not torch.equal(a, b)
    ja, wrist, images = runner.generate(3, 128)
    ja2, wrist2, images2 = runner.generate(3, 128)
    assert torch.equal(ja, ja2) and torch.equal(wrist, wrist2) and torch.equal(images, images2)
    assert ja.shape == (256, 22) and float(ja[:, 20:].abs().max()) == 0.0


def test_sharded_stream_equals_single_rank_stream(runner):
    from umetrack_amd.lib.common import metric_utils as mu
    from umetrack_amd.stream import chunk_plan
    from umetrack_amd import dist as udist
    total = 4 * 128 + 64                                   # ragged last chunk: one recording
    one = runner.run(total)
    assert one.frames_local == total and one.chunks_local == (0, 5)
    n_chunks, last = chunk_plan(total, 128, 64)
    assert (n_chunks, last) == (5, 64)
    parts = []
    for world in (2, 3):
        acc = torch.zeros_like(one.partials)
        frames = 0
        for rank in range(world):                          # what each rank of `world` would contribute
            c0, c1 = udist.shard_range(n_chunks, rank, world)
            for c in range(c0, c1):
                n = last if c == n_chunks - 1 else 128
                acc += runner.run_chunk(c, n)
                frames += n
        assert frames == total
        parts.append(acc)
    fin = mu.finalize_partials(one.partials)
    assert fin.n_total == 2 * total and fin.n_valid + one.invalid_local == 2 * total
    for acc in parts:
        f = mu.finalize_partials(acc)
        np.testing.assert_allclose([f.mpjpe, f.mpjpa, f.auc], [fin.mpjpe, fin.mpjpa, fin.auc], rtol=1e-12)
        assert (f.n_valid, f.n_total) == (fin.n_valid, fin.n_total)
    assert np.isfinite(fin.mpjpe) and fin.mpjpe > 0
