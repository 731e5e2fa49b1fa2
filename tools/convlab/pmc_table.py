#!/usr/bin/env python
"""Per-kernel averages of rocprofv3 --pmc passes over tools/convlab/convlab (one CSV per counter group).

    python tools/convlab/pmc_table.py gpurun_out/lab/pmc_*  -> table on stdout

SQ cycle counters (SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_*) are in quad-cycles summed over waves;
SQ_VALU_MFMA_BUSY_CYCLES is in cycles summed over the 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
import csv
import glob
import re
import sys
from collections import defaultdict


def main():
    rows = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    dur = defaultdict(lambda: [0.0, 0])
    seen = set()
    for pat in sys.argv[1:]:
        for path in sorted(glob.glob(pat + '/**/*counter_collection.csv', recursive=True)):
            with open(path) as f:
                for r in csv.DictReader(f):
                    name = re.sub(r'^void ', '', r['Kernel_Name']).replace('deva::(anonymous namespace)::', '')
                    m = re.match(r'(\w+)<([^>]*)>', name)
                    short = (m.group(1) + '<' + m.group(2).replace(' ', '') + '>') if m else re.sub(r'\(.*', '', name)
                    key = (short, r.get('Grid_Size', ''), r.get('LDS_Block_Size', ''))
                    c = rows[key][r['Counter_Name']]
                    c[0] += float(r['Counter_Value'])
                    c[1] += 1
                    k2 = (path, r['Dispatch_Id'])
                    if k2 not in seen:
                        seen.add(k2)
                        dur[key][0] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
                        dur[key][1] += 1
    for key in sorted(rows, key=lambda k: -dur[k][0] / max(dur[k][1], 1)):
        c = {k: v[0] / v[1] for k, v in rows[key].items()}
        us = dur[key][0] / max(dur[key][1], 1) / 1e3
        print(f'== {key[0]} grid {key[1]} : {us:.1f} us avg under counters, {dur[key][1]} dispatch-passes')
        gui = c.get('GRBM_GUI_ACTIVE', 0) / 8.0
        if gui:
            print(f'   GUI_ACTIVE/8 = {gui:.0f} cycles  -> clock {gui / us / 1e3:.2f} GHz')
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and gui:
            print(f'   MFMA util = {c["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024):.3f}')
        wc = c.get('SQ_WAVE_CYCLES', 0)
        for k in sorted(c):
            extra = ''
            if wc and k.startswith(('SQ_WAIT', 'SQ_ACTIVE_INST')):
                extra = f'  ({c[k] / wc:.3f} of wave cycles)'
            if k.startswith('SQ_INSTS_') and c.get('SQ_INSTS_MFMA'):
                extra = f'  ({c[k] / c["SQ_INSTS_MFMA"]:.2f} per MFMA)'
            print(f'   {k:32s} {c[k]:16.0f}{extra}')


if __name__ == '__main__':
    main()
