"""Writes tests/golden/oracle_snapshot.json: digests of what the CPU oracle frame loop produces on the 13-frame synthetic sequence
of tests/test_pipeline_oracle.py (poses, hash table, voxels, operation log).

This pins the ORACLE against accidental change between rounds.  It is NOT output of the reference: for that see reference_first_chunk.npz /
make_reference_golden.py next to this file and tests/test_ref_pin_cpu.py.

usage:  python tests/golden/make_oracle_snapshot.py        (from the repository root, after build())
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.test_pipeline_oracle import GOLDEN, run_oracle_sequence, snapshot  # noqa: E402

if __name__ == "__main__":
    op, _, _ = run_oracle_sequence()
    with open(GOLDEN, "w") as f:
        json.dump(snapshot(op), f, indent=1)
        f.write("\n")
    print("wrote", GOLDEN)
    op.scene.close()
