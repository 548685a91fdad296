"""Register / LDS / occupancy table of every kernel in a .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python profiles/kernel_resources.py faster_whisper_amd/csrc/dec_kernels.hip [filter]"""
import re, subprocess, sys, os
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = ["-mllvm", "-amdgpu-mfma-vgpr-form=1"] if os.path.basename(src) == "attn_enc.hip" else []
p = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *extra,
                    "-Rpass-analysis=kernel-resource-usage", "-c", os.path.abspath(src), "-o", "/dev/null"],
                   capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(src)) or ".")
cur = {}
rows = []
for line in p.stderr.splitlines():
    m = re.search(r"remark:\s+Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}; rows.append(cur); continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+)", line)
    if m and rows:
        cur[m.group(1).strip()] = m.group(2)
def dem(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip().split("(")[0]
    except Exception:
        return n
for r in rows:
    n = dem(r["name"])
    if flt and flt not in n: continue
    print(f"{n:70s} vgpr {r.get('VGPRs','?'):>4} agpr {r.get('AGPRs','?'):>4} occ {r.get('Occupancy','?'):>2} lds {r.get('LDS Size','?'):>6} scratch {r.get('ScratchSize','?')}")
