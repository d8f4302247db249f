"""`make -C failure_aware_vision_amd/csrc asm` -> a table of registers / scratch / LDS / occupancy per kernel instantiation
(from hipcc's -Rpass-analysis=kernel-resource-usage remarks): python tools/resource_table.py [resource_usage.txt] > profiles/rN_resource_usage.txt"""
import re, subprocess, sys, os
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "failure_aware_vision_amd", "lib", "asm", "resource_usage.txt")
rows, cur = [], None
for line in open(path):
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}; rows.append(cur); continue
    for key, pat in (("vgpr", r"VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"SGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("spill_s", r"SGPRs Spill: (\d+)"), ("spill_v", r"VGPRs Spill: (\d+)"),
                     ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None and key not in cur:
            cur[key] = int(m.group(1))
names = [r["name"] for r in rows]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
print(f"{'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'scratch':>8} {'spillV':>7} {'occ':>4} {'staticLDS':>10}  kernel")
for r, d in sorted(zip(rows, dem), key=lambda t: t[1]):
    d = re.sub(r"^void ", "", d); d = re.sub(r"\(.*\)$", "", d)
    print(f"{r.get('vgpr', 0):5d} {r.get('agpr', 0):5d} {r.get('sgpr', 0):5d} {r.get('scratch', 0):8d} {r.get('spill_v', 0):7d} {r.get('occ', 0):4d} {r.get('lds', 0):10d}  {d}")
