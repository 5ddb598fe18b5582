"""Per-kernel register / LDS / occupancy table of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python scripts/kernel_resources.py vllm_metax_amd/csrc/paged_attention.hip [substring-filter]"""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
src = Path(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", f"-I{ROOT / 'include'}", "-c", str(src),
       "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], {}
for line in out.splitlines():
    m = re.search(r"remark: (?:\S+: )?\s*([A-Za-z ]+?(?:\[[^\]]*\])?): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        if cur:
            rows.append(cur)
        cur = {"name": v}
    else:
        cur[k] = v
if cur:
    rows.append(cur)
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True,
                       text=True).stdout.splitlines()
print(f"{'VGPR':>5} {'AGPR':>5} {'spill':>5} {'scratch':>7} {'occ':>4} {'LDS':>7}  kernel")
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("mi355x::", "")
    if "_Accum" in n or n.startswith("_Z"):       # c++filt does not know DF16b / DF16_: decode the template list by hand
        m = re.match(r"_ZN6mi355x\d+([a-z0-9_]+?)I(.*?)EEv", r["name"])
        if m:
            args = m.group(2).replace("DF16b", "bf16,").replace("DF16_", "f16,").replace("Lb1E", "true,").replace("Lb0E", "false,")
            args = re.sub(r"Li(\d+)E", r"\1,", args).replace("h", "u8,").replace("f", "f32,")
            n = f"{m.group(1)}<{args.rstrip(',')}>"
    if flt and flt not in n:
        continue
    print(f"{r.get('VGPRs', '?'):>5} {r.get('AGPRs', '?'):>5} {r.get('VGPRs Spill', '?'):>5} "
          f"{r.get('ScratchSize [bytes/lane]', '?'):>7} {r.get('Occupancy [waves/SIMD]', '?'):>4} "
          f"{r.get('LDS Size [bytes/block]', '?'):>7}  {n}")
