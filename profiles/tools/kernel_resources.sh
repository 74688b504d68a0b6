#!/bin/bash
# Compiler resource report of every kernel (VGPRs, SGPRs, spills, scratch, occupancy, LDS) from
# -Rpass-analysis=kernel-resource-usage; runs on the build container (no GPU).  usage: kernel_resources.sh [extra hipcc flags]
cd "$(dirname "$0")/../../continiousenvironment_follower_leader_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -c ftl_abi.hip -o /tmp/ftl_abi_res.o \
    -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c '
import re, sys, subprocess
rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"remark: [^:]*:\d+:\d+: +(.*?) \[-Rpass", line) or re.search(r"remark: +(.*?) \[-Rpass", line)
    m = re.search(r"remark: +(.*?) \[-Rpass-analysis", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
def dem(n):
    try: return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip().split("(")[0]
    except Exception: return n
print("%-52s %6s %6s %7s %7s %8s %5s %7s" % ("kernel", "VGPRs", "SGPRs", "sgprSp", "vgprSp", "scratchB", "occ", "LDS B"))
for r in rows:
    print("%-52s %6s %6s %7s %7s %8s %5s %7s" % (dem(r["name"])[:52], r.get("VGPRs"), r.get("TotalSGPRs"), r.get("SGPRs Spill"), r.get("VGPRs Spill"),
          r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
'
