"""Register / occupancy report of a generated plugin's kernels (compiler remarks):
python tools/kernel_resources.py <plugin.hip>"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
out = f"/tmp/kres_{os.getpid()}.so"
res = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-I",
                      os.path.join(root, "directtrajectoryoptimization.jl_amd", "csrc"), "-Wno-unused-value", "-mllvm", "-amdgpu-mfma-vgpr-form",
                      "-Rpass-analysis=kernel-resource-usage", "-o", out, src], capture_output=True, text=True)
if os.path.exists(out):
    os.remove(out)
cur, rows = None, {}
for line in res.stderr.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
print(f"{'kernel':44s} VGPR AGPR spillV scratch occ   LDS")
for name, r in rows.items():
    short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    short = re.sub(r"<.*", "", short).replace("dto::", "").replace("void ", "")
    print(f"{short:44s} {r.get('VGPRs', '?'):>4s} {r.get('AGPRs', '?'):>4s} {r.get('VGPRs Spill', '?'):>6s} "
          f"{r.get('ScratchSize [bytes/lane]', '?'):>7s} {r.get('Occupancy [waves/SIMD]', '?'):>3s} {r.get('LDS Size [bytes/block]', '?'):>5s}")
