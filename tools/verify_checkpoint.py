#!/usr/bin/env python3
"""tools/verify_checkpoint.py <snapshot dir> [--subfolder hunyuan3d-dit-v2-0] [--variant fp16] [--steps 4] [--full] [--keys-only]

One command for the day a real Hunyuan3D-2 snapshot is reachable: key diff of the SAME files against the
oracle's module tree (strict=True) and the product's weight table, per-block deltas and max |residual| per block (decides the fp16
stream by data), N-step latents on the fp16 AND the fp32 stream, VAE output, grid logits, the texture loaders -- every figure at
the tolerance the parity tests use.  The implementation is test infrastructure (it drives oracle/) and lives in
tests/checkpoint_verify.py; this file is its command line.  Exit code 0 = inside every tolerance; JSON report on stdout."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import checkpoint_verify  # noqa: E402

if __name__ == "__main__":
    sys.exit(checkpoint_verify.main())
