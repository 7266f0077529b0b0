#!/usr/bin/env python3
"""The mutation fuzzer of tools/huff_fuzz.py pointed at the HOST emulation of the device entropy decoder
(tools/huff_emu): damaged scans, tables, restart intervals, truncations.  huff_emu's exit code is 0 only
when, image by image, the kernels either reproduce the host decoder's coefficients (status 0) or flag an
image the host decoder rejects too -- so a device decoder that accepts what the host rejects, or disagrees
on an accepted stream, fails here, without a GPU.  Usage: fuzz_emu.py [--n 200] [--seed 1] [--asan]"""
import argparse
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import huff_fuzz  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--exe", default=os.path.join(HERE, "huff_emu"))
    ap.add_argument("--per-run", type=int, default=8)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    base = huff_fuzz.seeds()
    done = bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        while done < args.n:
            paths = []
            for k in range(args.per_run):
                data = huff_fuzz.mutate(rng, base[int(rng.integers(0, len(base)))]) if done + k >= len(base) else bytes(base[done + k])
                p = os.path.join(tmp, f"m{k}.jpg")
                open(p, "wb").write(data)
                paths.append(p)
            env = dict(os.environ, JPEGBLK_CHUNK_BYTES=("64", "128")[int(rng.integers(0, 2))])
            r = subprocess.run([args.exe, "--quiet"] + paths, capture_output=True, text=True, timeout=600, env=env)
            done += len(paths)
            if r.returncode != 0:
                bad += 1
                print(r.stdout[-3000:], r.stderr[-3000:])
                keep = os.path.join(HERE, f"fuzz_fail_{bad}")
                os.makedirs(keep, exist_ok=True)
                for p in paths:
                    os.replace(p, os.path.join(keep, os.path.basename(p)))
                if bad >= 3:
                    break
    print(f"fuzz_emu: {done} streams, {bad} failing runs")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
