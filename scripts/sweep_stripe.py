"""Sweep the stripe-kernel plan (NW, SK) per Llama-3-8B decode shape: run on the GPU box.
Each configuration runs in a child process (the plan override is read from the environment)."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M = sys.argv[1] if len(sys.argv) > 1 else "64"
for name in ["qkv", "o", "gate_up", "down"]:
    best = None
    for nw in (2, 4):
        for sk in (1, 2, 4, 8, 16):
            env = dict(os.environ, MI355X_STRIPE_FORCE=f"{nw},{sk}")
            out = subprocess.run([sys.executable, os.path.join(root, "scripts/bench_gemm.py"), M, f"--only={name}"],
                                 env=env, capture_output=True, text=True).stdout
            us = [float(l.split(":")[1].split("us")[0]) for l in out.splitlines() if l.startswith("M=") and " K=" in l]
            if us:
                print(f"{name:8s} nw={nw} sk={sk:2d}: {us[0]:7.1f} us", flush=True)
                if best is None or us[0] < best[0]:
                    best = (us[0], nw, sk)
    print(f"BEST {name}: {best}", flush=True)
