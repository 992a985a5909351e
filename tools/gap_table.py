#!/usr/bin/env python3
"""Gap table of a train step: every row of a per-launch shape table (bench.py's eager single-stream leg,
profiles/*_per_launch_shapes_*.txt) priced at an ACHIEVABLE roof,

    roof_us = max(bytes / 6.3 TB/s, FLOP / 1.5 PFLOP/s) + 1.5 us   (measured copy rate, a well-fed MFMA loop, one launch boundary)

sorted by (measured - roof) x launches per step: where the step loses its time, same launches, no extra fusion.
usage: tools/gap_table.py profiles/r04_a_per_launch_shapes_r50.txt [--by-family]"""
import re
import sys

HBM, PF, BOUNDARY = 6.3e12, 1.5e15, 1.5e-6


def rows(path):
    pat = re.compile(r'^(\S+)\s+([\d.]+)/step\s+avg\s+([\d.]+) us\s+per-step\s+([\d.]+) ms\s+([\d.]+) MB\s+[\d.]+ TB/s\s+([\d.]+) GFLOP')
    for line in open(path):
        m = pat.match(line)
        if m:
            k, n, us, _, mb, gf = m.groups()
            yield k, float(n), float(us) * 1e-6, float(mb) * 1e6, float(gf) * 1e9


def main():
    path = sys.argv[1]
    fam = {}
    out = []
    for k, n, t, nb, fl in rows(path):
        roof = max(nb / HBM, fl / PF) + BOUNDARY
        gap = (t - roof) * n
        out.append((gap, k, n, t, roof, nb, fl))
        f = fam.setdefault(k, [0.0, 0.0, 0.0])
        f[0] += t * n; f[1] += roof * n; f[2] += n
    out.sort(reverse=True)
    print(f'{"kernel":22s} {"n/step":>6s} {"meas us":>8s} {"roof us":>8s} {"x":>5s} {"gap*n us":>9s} {"MB":>8s} {"GFLOP":>7s}  bound')
    for gap, k, n, t, roof, nb, fl in out:
        b = 'hbm' if nb / HBM >= fl / PF else 'mfma'
        print(f'{k:22s} {n:6.1f} {t * 1e6:8.1f} {roof * 1e6:8.1f} {t / roof:5.2f} {gap * 1e6:9.1f} {nb / 1e6:8.1f} {fl / 1e9:7.2f}  {b}')
    print()
    print(f'{"family":22s} {"launches":>8s} {"meas ms":>8s} {"roof ms":>8s} {"gap ms":>7s}')
    tm = tr = 0.0
    for k, (t, r, n) in sorted(fam.items(), key=lambda kv: -(kv[1][0] - kv[1][1])):
        print(f'{k:22s} {n:8.0f} {t * 1e3:8.3f} {r * 1e3:8.3f} {(t - r) * 1e3:7.3f}')
        tm += t; tr += r
    print(f'{"total":22s} {"":8s} {tm * 1e3:8.3f} {tr * 1e3:8.3f} {(tm - tr) * 1e3:7.3f}')


if __name__ == '__main__':
    main()
