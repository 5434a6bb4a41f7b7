#!/bin/bash
# Audit of tail2_kernel's ISA (encoder_tail2.hip keeps its accumulators in AGPRs owned by inline assembly):
# no spills / scratch, no compiler-generated AGPR or MFMA instruction, no compiler vmcnt(0) inside the loops.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I $ROOT/memex_amd/csrc -I $ROOT/scripts -S --cuda-device-only $ROOT/scripts/encoder_tail2.hip -o $T/t2.s -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "VGPRs:|AGPRs:|Spill|ScratchSize" | sed 's/.*remark: *//'
python3 - $T/t2.s <<'PY'
import re, sys, statistics
lines = open(sys.argv[1]).read().split('\n')
start = [i for i, l in enumerate(lines) if l.startswith('_ZN2mx12tail2_kernel')][0]
end = [i for i, l in enumerate(lines) if 's_endpgm' in l and i > start][0]
inasm = False; bad = []; waits = {}; mf = []
for i in range(start, end):
    l = lines[i]
    if '#ASMSTART' in l: inasm = True; continue
    if '#ASMEND' in l: inasm = False; continue
    code = l.split(';')[0]
    if not inasm and re.search(r'v_accvgpr|v_mfma|\ba\[\d+|\ba\d+\b|scratch_', code): bad.append((i, l.strip()))
    if 's_waitcnt' in code and 'vmcnt' in code: waits[(code.strip(), 'asm' if inasm else 'compiler')] = waits.get((code.strip(), 'asm' if inasm else 'compiler'), 0) + 1
    if 'v_mfma' in code: mf.append(i)
def ninstr(a, b): return sum(1 for l in lines[a:b] if l.strip() and not l.strip().startswith(';') and not l.strip().startswith('.') and not l.strip().endswith(':'))
gaps = [ninstr(mf[j], mf[j + 1]) - 1 for j in range(len(mf) - 1)]
steady = sorted(gaps)[: int(len(gaps) * 0.95)]
print('mfma', len(mf), '| compiler-side AGPR/MFMA/scratch instructions:', len(bad), bad[:3])
print('vmcnt waits:', waits)
print('instructions between consecutive MFMAs: median %d, mean of the lower 95%% %.1f, max %d' % (statistics.median(gaps), statistics.mean(steady), max(gaps)))
PY
rm -rf $T
