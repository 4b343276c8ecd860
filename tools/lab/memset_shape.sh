cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/msp -o t --output-format csv -- $GRAFT_REPO_ROOT/tools/lab/_build/fill_pattern 4 30000 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/msp/**/*kernel_trace.csv',recursive=True)[0]
seen=set()
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    if 'fill' in n.lower() or 'memset' in n.lower():
        key=(n,r['Grid_Size_X'],r['Workgroup_Size_X'])
        if key not in seen:
            seen.add(key); print(n[:80], 'grid',r['Grid_Size_X'],r.get('Grid_Size_Y'),'wg',r['Workgroup_Size_X'], 'dur', (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, 'lds', r.get('LDS_Block_Size'), 'vgpr', r.get('VGPR_Count'))
PY
