"""Not a test: prints occupancy / issue figures from the counter CSVs tools/prof_occ.sh leaves under gpurun_out/occ/."""
import collections
import csv
import glob

for f in sorted(glob.glob('gpurun_out/occ/*counter_collection.csv')):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(list)
    for r in rows:
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
    m = {k: sum(v) / len(v) for k, v in agg.items()}
    gui = m['GRBM_GUI_ACTIVE'] / 8
    print(f.split('/')[-1].replace('_counter_collection.csv', ''), 'lds', rows[0]['LDS_Block_Size'], 'cycles/XCD %.3g' % gui,
          'waves/SIMD %.2f' % (m['SQ_WAVE_CYCLES'] * 4 / (gui * 1024)), 'VALU busy %.2f' % (m['SQ_ACTIVE_INST_VALU'] * 4 / (gui * 1024)),
          'VALU insts/wave %.0f' % (m['SQ_INSTS_VALU'] / m['SQ_WAVES']), 'LDS busy %.2f' % (m['SQ_LDS_IDX_ACTIVE'] / (gui * 256)))
