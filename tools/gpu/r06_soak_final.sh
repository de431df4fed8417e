#!/bin/bash
# round 6, final sources: the full-size step held to itself (lr = 0), eagerly and replayed, alone and next to a second process
set -u
O=gpurun_out/r06_soak_final; mkdir -p $O
( MODE=eager STEPS=400 PROCS=1 timeout 900 python tools/step_soak_determinism.py ) > $O/eager_1.txt 2>&1; grep -E "proc " $O/eager_1.txt | cut -c1-200
( MODE=graph STEPS=1500 PROCS=1 PCHK=0 timeout 900 python tools/step_soak_determinism.py ) > $O/graph_1.txt 2>&1; grep -E "proc " $O/graph_1.txt | cut -c1-200
( MODE=eager STEPS=300 PROCS=2 timeout 1200 python tools/step_soak_determinism.py ) > $O/eager_2.txt 2>&1; grep -E "proc " $O/eager_2.txt | cut -c1-200
( MODEL=swin_t BITS=3 MODE=eager STEPS=200 PROCS=1 timeout 1200 python tools/step_soak_determinism.py ) > $O/swin_eager_1.txt 2>&1; grep -E "proc " $O/swin_eager_1.txt | cut -c1-200
( MODEL=swin_t BITS=3 MODE=graph STEPS=500 PROCS=1 PCHK=0 timeout 1200 python tools/step_soak_determinism.py ) > $O/swin_graph_1.txt 2>&1; grep -E "proc " $O/swin_graph_1.txt | cut -c1-200
for i in 1 2 3; do timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-recipe-line > $O/bench_$i.json 2>/dev/null; python -c "import json; d=json.loads(open('$O/bench_$i.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'])"; done
