# VGPR / AGPR / scratch / occupancy of every gemm_nt / gemm_tn instantiation (compiler view; no GPU needed).
# usage: bash tests/tools/kernel_resources.sh [pattern]     (measurement tool)
cd "$(dirname "$0")/../../stil_tta_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -o /tmp/stil_res.o stil_hip.hip -Rpass-analysis=kernel-resource-usage 2>&1 \
 | python3 -c "
import sys,re,subprocess
pat=sys.argv[1] if len(sys.argv)>1 else 'gemm_'
cur=None;rows=[]
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m:
        cur={'name':m.group(1)};rows.append(cur);continue
    for k in ('VGPRs','AGPRs','ScratchSize \[bytes/lane\]','Occupancy \[waves/SIMD\]','LDS Size \[bytes/block\]','SGPRs'):
        m=re.search(r'remark:\s+'+k+r': (\d+)',l)
        if m and cur is not None: cur[k.split(' ')[0]]=int(m.group(1))
names=subprocess.run(['c++filt']+[r['name'] for r in rows],capture_output=True,text=True).stdout.split('\n')
for r,n in zip(rows,names):
    if pat in n: print(f\"{r.get('VGPRs',0):4d} v {r.get('AGPRs',0):3d} a {r.get('SGPRs',0):3d} s scratch {r.get('ScratchSize',0):3d} occ {r.get('Occupancy',0)}  {n[:100]}\")
" "$1"
