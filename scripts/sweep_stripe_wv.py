"""Sweep the K split of the decode stripe kernel per Llama-3-8B shape for 8- and 16-wave workgroups (GPU box)."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M = sys.argv[1] if len(sys.argv) > 1 else "64"
for name in ["qkv", "o", "gate_up", "down"]:
    for wv in (8, 16):
        best = None
        for sk in (1, 2, 3, 4, 5, 6, 7, 8):
            if name == "gate_up" and sk > 2:
                continue
            env = dict(os.environ, MI355X_STRIPE_FORCE=f"2,{sk}", MI355X_STRIPE_WV=str(wv))
            out = subprocess.run([sys.executable, os.path.join(root, "scripts/bench_gemm.py"), M, f"--only={name}"],
                                 env=env, capture_output=True, text=True).stdout
            us = [float(l.split(":")[1].split("us")[0]) for l in out.splitlines() if l.startswith("M=") and " K=" in l]
            if us:
                print(f"{name:8s} wv={wv:2d} sk={sk:2d}: {us[0]:7.1f} us", flush=True)
                if best is None or us[0] < best[0]:
                    best = (us[0], sk)
        print(f"BEST {name} wv={wv}: {best}", flush=True)
