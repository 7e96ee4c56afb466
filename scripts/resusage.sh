#!/bin/bash
# usage: scripts/resusage.sh tinynerf_amd/csrc/file.hip [extra hipcc flags]
# one line per kernel: VGPRs / AGPRs / SGPRs, spills, scratch bytes per lane, occupancy (hipcc -Rpass-analysis=kernel-resource-usage)
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fhip-fp32-correctly-rounded-divide-sqrt "$@" -c $f -o /tmp/_ru.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re,subprocess
cur={}
rows=[]
for ln in sys.stdin:
    m=re.search(r'remark:\s+(.*?) \[-Rpass',ln)
    if not m: continue
    t=m.group(1).strip()
    if t.startswith('Function Name:') or t.startswith('Name:'):
        if cur: rows.append(cur)
        cur={'name':t.split(':',1)[1].strip()}
    elif ':' in t:
        k,v=t.split(':',1); cur[k.strip()]=v.strip()
if cur: rows.append(cur)
for r in rows:
    try: nm=subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt',r['name']],capture_output=True,text=True).stdout.strip()
    except Exception: nm=r['name']
    nm=nm.replace('(anonymous namespace)::','').split('(')[0].replace('void ','')
    print('%-62s v%4s a%4s s%4s  vspill %4s sspill %4s scratch %5s occ %s lds %s'%(nm[:62],r.get('VGPRs'),r.get('AGPRs'),r.get('TotalSGPRs'),r.get('VGPRs Spill', r.get('VGPR Spill')),r.get('SGPRs Spill', r.get('SGPR Spill')),r.get('ScratchSize [bytes/lane]'),r.get('Occupancy [waves/SIMD]'),r.get('LDS Size [bytes/block]')))
"
