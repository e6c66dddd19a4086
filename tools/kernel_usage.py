"""Builds libmirhi.so with -Rpass-analysis=kernel-resource-usage and prints one line per kernel:
VGPRs, SGPRs, spills, scratch, occupancy, LDS.  usage: python tools/kernel_usage.py"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(ROOT, "renderer-rs_amd", "build.py"), "--usage", "--force"], capture_output=True, text=True)
text = out.stdout + out.stderr
cur = None; rows = {}
for line in text.splitlines():
    m = re.search(r"remark:\s+(Function Name|VGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]):\s*(\S+)", line)
    if not m: continue
    k, v = m.groups()
    if k == "Function Name":
        cur = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip() or v
        cur = re.sub(r"\(.*", "", cur).replace("void mirhi::", "").replace("mirhi::", "")
        rows[cur] = {}
    elif cur: rows[cur][k.split(" [")[0]] = v
print(f"{'kernel':38s} VGPR SGPR vspill sspill scratch occ   LDS")
for k, r in rows.items():
    print(f"{k:38s} {r.get('VGPRs','?'):>4s} {r.get('TotalSGPRs','?'):>4s} {r.get('VGPRs Spill','?'):>6s} {r.get('SGPRs Spill','?'):>6s} {r.get('ScratchSize','?'):>7s} {r.get('Occupancy','?'):>3s} {r.get('LDS Size','?'):>6s}")
bad = [k for k, r in rows.items() if r.get("ScratchSize", "0") != "0" or r.get("VGPRs Spill", "0") != "0"]
print("kernels with scratch or VGPR spills:", bad or "none")
