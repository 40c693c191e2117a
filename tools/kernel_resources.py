"""Register / scratch / occupancy table of every kernel in h2gcn_capi.hip (compiler remarks; no GPU needed).
usage: python tools/kernel_resources.py [--used-in profile.txt ...] [extra hipcc flags]
  --used-in: print only the spmm_hops_kernel instantiations whose (demangled) names appear in the given rocprofv3-derived
             profile texts (profiles/*_epoch_*.txt, *_train_step_*.txt, *_kernel_stats.csv ...), with the ms / calls columns of
             the line they appear in -- i.e. the resources of the kernels a workload REALLY dispatched."""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
used = {}
argv = sys.argv[1:]
while "--used-in" in argv:
    i = argv.index("--used-in")
    j = i + 1
    while j < len(argv) and not argv[j].startswith("-"):
        for line in Path(argv[j]).read_text().splitlines():
            m = re.search(r"spmm_hops_kernel<(\d+), (\d+), (true|false), (true|false), (true|false), (true|false), (true|false), (true|false), (\d+)(?:, (true|false))?>", line)
            if m:
                key = tuple("1" if v == "true" else "0" if v in ("false", None) else v for v in m.groups())
                used.setdefault(key, []).append(f"{Path(argv[j]).name}: {' '.join(line.split()[-2:])}")
        j += 1
    del argv[i:j]
sys.argv[1:] = argv
txt = ""
for unit in ("h2gcn_capi.hip", "spmm_short.hip"):      # the translation units that hold spmm_hops_kernel instantiations
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include", "--offload-arch=gfx950",
           "--cuda-device-only", "-c", "-Rpass-analysis=kernel-resource-usage", str(ROOT / "h2gcn_amd/csrc" / unit),
           "-o", "/tmp/_kres.o"] + sys.argv[1:]
    txt += subprocess.run(cmd, capture_output=True, text=True).stderr
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0].strip()

    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    m = re.search(r"spmm_hops_kernelILi(\d+)ELi(\d+)ELb(\d)ELb(\d)ELb(\d)ELb(\d)ELb(\d)ELb(\d)ELi(\d+)ELb(\d)E", name)
    if used and not (m and m.groups() in used):
        continue
    where = ("   <- " + "; ".join(used[m.groups()])) if used and m else ""
    if m:
        v, lpr, ex, su, o32, pipe, sh, epi, fb, lists = m.groups()
        name = f"spmm<VEC{v} LPR{lpr:>2s} {'EXACT' if ex=='1' else 'tiled'} {'SUM' if su=='1' else 'fwd'} {'off32' if o32=='1' else 'off64'}" \
               f"{' PIPE' if pipe=='1' else ''}{' SHORT' if sh=='1' else ''}{' LISTS' if lists=='1' else ''}{' GEN' if epi=='1' else ''} FB{fb}>"
    scratch, occ = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")
    print(f"{name:64s} vgpr {g('VGPRs'):3d} sgpr {g('SGPRs'):3d} scratch {scratch:3d} occupancy {occ}{where}")
