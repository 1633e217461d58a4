"""CPU: the oracle frame loop against the golden vectors the REFERENCE ITSELF produced (tests/golden/reference_first_chunk.npz; how they were
made: tests/golden/make_reference_golden.py, contents: tests/golden_ref.py).  The GPU twin, tests/test_golden_ref_gpu.py, holds the product to the
same fixture on the MI355X box, where the reference does not exist."""
import numpy as np

from tests import golden_ref as G


def test_fixture_is_what_the_generator_describes():
    g = np.load(G.PATH)
    assert g["poses"].shape == (G.N, 4, 4) and g["valid"].all()
    assert np.array_equal(g["poses"][0], np.eye(4, dtype=np.float32))              # the first frame defines the world
    assert len(g["corr"]) > 500 and len(g["keys0"]) == len(g["desc0_sum"]) > 100 and len(g["blocks"]) == len(g["block_crc"]) > 300
    assert (g["corr"]["imgIdx_i"] < g["corr"]["imgIdx_j"]).all() and g["corr"]["imgIdx_j"].max() == G.N - 1


def test_oracle_frame_loop_reproduces_the_reference_bit_for_bit(oracle):
    """Tracked poses, the chunk's correspondences, the key points of frame 0, and the volume (block set, every voxel byte, free list)."""
    from tests.oracle_pipeline import OraclePipeline
    g = np.load(G.PATH)
    gas, gbs = G.params()
    frames, K = G.stream()
    op = OraclePipeline(gas, gbs, G.W, G.H, K)
    for d, c, _, _ in frames:
        op.process_frame(d, c)
    traj = op.integrated_trajectory()
    keys0, dsum0 = G.sorted_keys(op.local.keys[0], op.local.descs[0])
    blocks, crc, free = G.volume_digest(op.scene.hash(), op.scene.voxels(), op.scene.heap_counter())
    G.check(g, traj, np.isfinite(np.asarray(traj)[:, 0, 0]), op.local.corr, keys0, dsum0, blocks, crc, free, "oracle")
