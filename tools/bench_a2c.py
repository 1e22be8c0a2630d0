#!/usr/bin/env python3
"""End-to-end synchronous A2C on the HIP env (BASELINE configs[2] on one GPU, configs[3] with --gpus 8): a thin front end of
`bench.py --mode a2c`, kept under its round-1 name.

  python tools/bench_a2c.py [--envs 8192] [--rollouts 6] [--gpus N]
Prints bench.py's JSON line: env-steps/s including policy inference, sampling, env step, the update and its gradient all-reduce."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=8192)
    ap.add_argument("--rollouts", type=int, default=6)
    ap.add_argument("--gpus", type=int, default=int(os.environ.get("WORLD_SIZE", "1")))
    a = ap.parse_args()
    import bench

    bench.main(["--mode", "a2c", "--envs", str(a.envs), "--steps", str(50 * a.rollouts if a.rollouts != 40 else 1950), "--gpus", str(a.gpus)])


if __name__ == "__main__":
    main()
