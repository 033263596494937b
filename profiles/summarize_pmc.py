"""Condense rocprofv3 --pmc counter_collection.csv files into a per-kernel summary.

    python profiles/summarize_pmc.py <fetch_dir> <write_dir> <out.json>

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE tallies 128-byte read requests at
64 bytes, i.e. reports half the bytes of coalesced streaming reads; the factor 2 was checked here
on kernels of this library with exactly known traffic (k_points_step reads 4 x 24 MB and reports
46 910 KiB; k_gauge_points reads 24 MB and reports 11 730 KiB).  WRITE_SIZE matched the known
byte counts as is (k_gauge_points 23 437.5 KiB for 24 MB)."""
import collections
import csv
import glob
import json
import sys


def per_kernel(dirname, counter):
    f = glob.glob(dirname + '/**/*counter_collection.csv', recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == counter:
            agg[r['Kernel_Name']].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


def main():
    fetch, nf = per_kernel(sys.argv[1], 'FETCH_SIZE')
    write, _ = per_kernel(sys.argv[2], 'WRITE_SIZE')
    out = {}
    for k in fetch:
        if 'ba::' not in k:
            continue
        rd = fetch[k] * 1024 * 2.0
        wr = write.get(k, 0.0) * 1024
        out[k.split('(')[0]] = {'launches': nf[k], 'FETCH_SIZE_KiB_avg': fetch[k], 'WRITE_SIZE_KiB_avg': write.get(k),
                                'read_bytes_corrected': rd, 'write_bytes': wr, 'hbm_bytes_per_launch': rd + wr}
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
    for k, v in out.items():
        print(f"{k:40s} {v['hbm_bytes_per_launch'] / 1e6:10.2f} MB/launch")


if __name__ == '__main__':
    main()
