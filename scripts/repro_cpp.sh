#!/bin/bash
# is fmmap_amd's result a function of its input?  the synthetic workload several times, with and without some of the extension's routes
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p /tmp/repro
i=0
for flags in 0 65536 0 65536 0 16384 0 2 0 32768 0 0; do
  i=$((i+1))
  $R/nvbio-gpl_amd/lib/fmmap_amd --synthetic --genome-len 3e9 --n-reads 1e7 --read-len 150 --steps 2 --algo-flags $flags --out /tmp/repro/out_$i.bin 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('run $i flags $flags', d['checksum'], d['ms_per_step'])
    elif l.strip(): print(l.strip()[:200])
"
done
python3 - <<'PY'
import numpy as np, glob, collections
R = 10_000_000
outs = {}
for f in sorted(glob.glob('/tmp/repro/out_*.bin'), key=lambda x: int(x.split('_')[-1].split('.')[0])):
    raw = np.fromfile(f, dtype=np.uint8)
    outs[f] = (raw[:4*R].view(np.int32).copy(), raw[4*R:12*R].view(np.int64).copy(), raw[12*R:13*R].copy())
ref = outs['/tmp/repro/out_2.bin']
for f, (s, p, r) in outs.items():
    d = np.nonzero((s != ref[0]) | (p != ref[1]) | (r != ref[2]))[0]
    msg = ''
    if len(d):
        c = collections.Counter(zip(s[d].tolist(), ref[0][d].tolist()))
        msg = str(c.most_common(6)) + ' first ' + str(d[:6].tolist())
    print(f, 'differs from run 2 in', len(d), 'reads', msg)
PY
