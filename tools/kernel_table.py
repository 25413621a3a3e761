"""Per-kernel table of one forward from two rocprofv3 --pmc passes (tools/profile_session.sh layout): python tools/kernel_table.py gpurun_out/<tag>"""
import csv, collections, sys
d = sys.argv[1]
def load(path):
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        k = int(r['Dispatch_Id'])
        e = rows.setdefault(k, {'name': r['Kernel_Name'], 'grid': int(r['Grid_Size']), 'wg': int(r['Workgroup_Size']), 'vgpr': int(r['VGPR_Count']), 'dur': int(r['End_Timestamp'])-int(r['Start_Timestamp'])})
        e[r['Counter_Name']] = float(r['Counter_Value'])
    return rows
a = load(d + '/pmc_sqa/a_counter_collection.csv'); b = load(d + '/pmc_sqb/b_counter_collection.csv')
def last_fwd(rows):
    keys = list(rows.keys())
    idx = [i for i,k in enumerate(keys) if 'stem_block' in rows[k]['name']]
    return [rows[k] for k in keys[idx[-2]:idx[-1]]]
tot = 0
for x, y in zip(last_fwd(a), last_fwd(b)):
    w = x.get('SQ_WAVES', 0)
    if not w: continue
    nm = x['name'].replace('vbt::','').replace('void ','')[:50]
    tot += y.get('SQ_INSTS_VALU',0)
    print(f"{nm:50s} v{x['vgpr']:3d} dur{x['dur']/1000:6.1f} waves{w:7.0f} VALU/w{y.get('SQ_INSTS_VALU',0)/w:6.0f} MFMA/w{y.get('SQ_INSTS_MFMA',0)/w:5.0f} LDS/w{y.get('SQ_INSTS_LDS',0)/w:5.0f} conf{y.get('SQ_LDS_BANK_CONFLICT',0)/max(1,y.get('SQ_LDS_IDX_ACTIVE',1)):.2f} act_valu{x.get('SQ_ACTIVE_INST_VALU',0)/max(1,x.get('SQ_WAVE_CYCLES',1)):.2f} wait{x.get('SQ_WAIT_ANY',0)/max(1,x.get('SQ_WAVE_CYCLES',1)):.2f} MVALU{y.get('SQ_INSTS_VALU',0)/1e6:6.2f}")
print('total VALU M', tot/1e6)
