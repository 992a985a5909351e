#!/usr/bin/env python3
"""Per train step of a rocprofv3 kernel trace (tools/gpu_timeline.sh): where the step ends.  A step = from one pack_weights launch
to the next; for the last full steps: step time, busy time of each queue, and the tail - what runs between the end of the main
chain's last backward kernel and the start of sgd_kernel (only weight-gradient kernels of the side stream are left then)."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
key_s = 'Start_Timestamp' if 'Start_Timestamp' in rows[0] else 'start_timestamp'
key_e = 'End_Timestamp' if 'End_Timestamp' in rows[0] else 'end_timestamp'
name_k = 'Kernel_Name' if 'Kernel_Name' in rows[0] else 'kernel_name'
q_k = next(k for k in rows[0] if k.lower() in ('queue_id', 'stream_id'))
ev = sorted(((int(r[key_s]), int(r[key_e]), r[name_k], r[q_k]) for r in rows), key=lambda e: e[0])
packs = [i for i, e in enumerate(ev) if e[2].startswith('pack_weights_kernel')]
print(f'{len(ev)} dispatches, {len(packs)} pack_weights launches, queues: {sorted(set(e[3] for e in ev))}')
for a, b in list(zip(packs, packs[1:]))[-4:]:
    step = ev[a:b]
    t0, t1 = step[0][0], ev[b][0]
    sgd = [e for e in step if e[2].startswith('sgd_kernel')]
    if not sgd:
        continue
    ts = sgd[0][0]
    qmain = step[0][3]
    main_before = [e for e in step if e[3] == qmain and e[1] <= ts]
    last_main = max(main_before, key=lambda e: e[1])
    tail = [e for e in step if e[0] < ts and e[1] > last_main[1] and e[3] != qmain]
    busy = defaultdict(int)
    for e in step:
        busy[e[3]] += e[1] - e[0]
    print(f'step {1e-6 * (t1 - t0):.3f} ms; busy per queue {{{", ".join(f"{q}: {1e-6 * v:.3f}" for q, v in busy.items())}}}; '
          f'last main kernel before sgd: {last_main[2][:40]} ends {1e-3 * (ts - last_main[1]):.1f} us before sgd starts')
    mq = sorted((e for e in step if e[3] == qmain), key=lambda e: e[0])
    gaps = sorted(((b[0] - a[1], a, b) for a, b in zip(mq, mq[1:])), key=lambda g: -g[0])
    tot_gap = sum(g[0] for g in gaps if g[0] > 0)
    print(f'    main queue: {len(mq)} launches, idle between launches {1e-3 * tot_gap:.0f} us in total; the largest gaps:')
    for g, a, b in gaps[:6]:
        print(f'      {1e-3 * g:7.1f} us between {a[2][:44]:44s} and {b[2][:44]}')
        for e in step:      # what the other queues ran meanwhile
            if e[3] != qmain and e[0] < b[0] and e[1] > a[1]:
                print(f'                 meanwhile on queue {e[3]}: {e[2][:50]:50s} {1e-3 * (e[1] - e[0]):7.1f} us, ends {1e-3 * (b[0] - e[1]):+6.1f} us before the gap closes')
    for e in tail:
        print(f'    tail: {e[2][:60]:60s} {1e-3 * (e[1] - e[0]):7.1f} us  (starts {1e-3 * (e[0] - last_main[1]):+7.1f} us after the main chain ended)')
