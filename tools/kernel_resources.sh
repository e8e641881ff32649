#!/bin/bash
# per-kernel register / scratch / LDS table of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage), e.g.
#   tools/kernel_resources.sh kivi_amd/csrc/kivi_mf.hip
f=$1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math $KIVI_EXTRA_FLAGS -c "$f" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re,sys,subprocess
rows=[];cur=None
for l in sys.stdin:
    m=re.search(r"Function Name: (\S+)",l)
    if m:
        name=subprocess.run(["/usr/bin/c++filt",m.group(1)],capture_output=True,text=True).stdout.strip()
        name=re.sub(r"\(anonymous namespace\)::","",name); name=re.sub(r"\(.*","",name); name=name.replace("void ","")
        cur={"name":name}; rows.append(cur); continue
    for k,pat in (("sgpr",r"TotalSGPRs: (\d+)"),("vgpr",r" VGPRs: (\d+)"),("agpr",r"AGPRs: (\d+)"),("scratch",r"ScratchSize \[bytes/lane\]: (\d+)"),("occ",r"Occupancy \[waves/SIMD\]: (\d+)"),("lds",r"LDS Size \[bytes/block\]: (\d+)"),("vspill",r"VGPRs Spill: (\d+)")):
        m=re.search(pat,l)
        if m and cur is not None: cur[k]=int(m.group(1))
print("%-60s %5s %5s %5s %7s %4s %6s"%("kernel","vgpr","agpr","sgpr","scratch","occ","lds"))
for r in rows: print("%-60s %5d %5d %5d %7d %4d %6d"%(r["name"][:60],r.get("vgpr",0),r.get("agpr",0),r.get("sgpr",0),r.get("scratch",0),r.get("occ",0),r.get("lds",0)))
'
