#!/usr/bin/env python3
"""Summarise the rocprofv3 PMC passes of the batched lookahead (tools/bench_lookahead.py K): float64 matrix-core
flops executed (SQ_INSTS_VALU_MFMA_MOPS_F64 x 512), MFMA busy cycles, HBM bytes (FETCH_SIZE x 2 on gfx950 +
WRITE_SIZE, per the guide's HBM section) over ALL kernels of the run, and the rates against the time of an
unprofiled run of the same command.
usage: pmc_lookahead.py <unprofiled.json> <sq_counter_collection.csv> <fetch.csv> <write.csv>"""
import collections
import csv
import json
import sys

F64_MFMA_PEAK_TFLOPS = 78.6


def totals(path):
    tot = collections.defaultdict(float)
    per_kernel = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = set()
    for r in csv.DictReader(open(path)):
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0]
        tot[r['Counter_Name']] += float(r['Counter_Value'])
        per_kernel[name][r['Counter_Name']] += float(r['Counter_Value'])
        disp.add(r.get('Dispatch_Id'))
    return tot, per_kernel, len(disp)


def stamp():
    """When and on which build the counters were collected (bench.py quotes both beside anything it takes from here)."""
    import datetime
    import hashlib
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for name in ('libapmf_hip.so', 'libapmf_dense.so', 'libapmf_comm.so'):
        try:
            with open(os.path.join(root, 'active_matrix_factorization_amd', name), 'rb') as f:
                h.update(f.read())
        except OSError:
            h.update(b'missing:' + name.encode())
    return {'collected': datetime.datetime.utcnow().strftime('%Y-%m-%dT%H:%MZ'), 'build_id': h.hexdigest()[:16]}


def main():
    run = json.load(open(sys.argv[1]))
    sq, sq_k, ndisp = totals(sys.argv[2])
    fetch, _, _ = totals(sys.argv[3])
    write, _, _ = totals(sys.argv[4])
    flops = sq['SQ_INSTS_VALU_MFMA_MOPS_F64'] * 512.
    secs = run['seconds']
    hbm = (2. * fetch['FETCH_SIZE'] + write['WRITE_SIZE']) * 1024.
    top = sorted(sq_k.items(), key=lambda kv: -kv[1].get('SQ_INSTS_VALU_MFMA_MOPS_F64', 0))[:8]
    out = {'command': 'tools/bench_lookahead.py %d' % run['candidates'], 'hypotheses': run['hypotheses'],
           'seconds_unprofiled': secs, 'hypotheses_per_s': run['hypotheses_per_s'], 'dispatches': ndisp,
           'mfma_f64_flops': flops, 'mfma_f64_tflops': flops / secs / 1e12,
           'mfma_f64_frac_of_peak': flops / secs / 1e12 / F64_MFMA_PEAK_TFLOPS, 'peak_tflops': F64_MFMA_PEAK_TFLOPS,
           'SQ_VALU_MFMA_BUSY_CYCLES': sq.get('SQ_VALU_MFMA_BUSY_CYCLES'), 'SQ_BUSY_CYCLES': sq.get('SQ_BUSY_CYCLES'),
           'hbm_bytes': hbm, 'hbm_GBs': hbm / secs / 1e9, 'hbm_frac_of_8TBs': hbm / secs / 8e12,
           'bytes_per_hypothesis': hbm / run['hypotheses'], 'flops_per_hypothesis': flops / run['hypotheses'],
           'correction': 'FETCH_SIZE x2 (gfx950, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported; MOPS x 512 flops',
           'mfma_flops_by_kernel': {k: v.get('SQ_INSTS_VALU_MFMA_MOPS_F64', 0) * 512. for k, v in top}}
    out.update(stamp())
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
