"""Register / scratch / occupancy table of every kernel in h2gcn_capi.hip (compiler remarks; no GPU needed).
usage: python tools/kernel_resources.py [extra hipcc flags]"""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include", "--offload-arch=gfx950",
       "--cuda-device-only", "-c", "-Rpass-analysis=kernel-resource-usage", str(ROOT / "h2gcn_amd/csrc/h2gcn_capi.hip"),
       "-o", "/tmp/_kres.o"] + sys.argv[1:]
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0].strip()

    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    m = re.search(r"spmm_hops_kernelILi(\d+)ELi(\d+)ELb(\d)ELb(\d)ELb(\d)ELb(\d)ELb(\d)ELb(\d)ELi(\d+)E", name)
    if m:
        v, lpr, ex, su, o32, pipe, sh, epi, fb = m.groups()
        name = f"spmm<VEC{v} LPR{lpr:>2s} {'EXACT' if ex=='1' else 'tiled'} {'SUM' if su=='1' else 'fwd'} {'off32' if o32=='1' else 'off64'}" \
               f"{' PIPE' if pipe=='1' else ''}{' SHORT' if sh=='1' else ''}{' GEN' if epi=='1' else ''} FB{fb}>"
    scratch, occ = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")
    print(f"{name:64s} vgpr {g('VGPRs'):3d} sgpr {g('SGPRs'):3d} scratch {scratch:3d} occupancy {occ}")
