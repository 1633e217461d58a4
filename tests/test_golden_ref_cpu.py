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


def test_oracle_frame_loop_follows_the_reference_through_global_solves(oracle):
    """tests/golden/reference_stream_121.npz: the emulated reference on 121 frames 0.2 degrees apart at 320x240 - twelve chunks, eleven global solves,
    re-integration scheduling throughout; well-conditioned (every chunk solved from a good guess, the raw-match cap not reached), so the 1 mm bar
    of north_star and 5e-4 per pose can be asserted: online poses, final trajectory, key frames, the scheduled TSDF operations."""
    import os
    from tests.oracle_pipeline import OraclePipeline
    m = G.stream_fixture()
    if not os.path.exists(m.PATH):
        import pytest
        pytest.skip("fixture not generated yet")
    g = np.load(m.PATH)
    frames, K = m.stream()
    op = OraclePipeline(*m.params(), m.W, m.H, K)
    op._integrate_orig = op._integrate
    online = np.full((m.NF, 4, 4), -np.inf, np.float32)
    ops = []

    def log_only(frame, T, de):          # no volume: the schedule is what is compared
        ops.append((1 if de else 0, frame))
    op._integrate = log_only
    for i, (d, c, _, _) in enumerate(frames):
        op.process_frame(d, c)
        if op.last_valid:
            online[i] = op.cur_T[op.last_processed]
    for _ in range(m.TAIL):
        op.process_end_of_sequence()
    # the oracle logs a re-integration as de + in of the same frame: fold into the reference's (kind 2, frame)
    folded = []
    k = 0
    while k < len(ops):
        if ops[k][0] == 1 and k + 1 < len(ops) and ops[k + 1] == (0, ops[k][1]):
            folded.append((2, ops[k][1])); k += 2
        else:
            folded.append(ops[k]); k += 1
    final = np.full((m.NF, 4, 4), -np.inf, np.float32)
    n = min(op.num_complete, m.NF)
    final[:n] = np.asarray(op.complete[:n], np.float32)
    G.check_stream(g, online, final, op.glob.num_images, folded, frames, "oracle")
