#!/bin/bash
# names of the vendor-library GEMM kernels on the C2 shapes (diagnostic)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/libk
timeout 240 rocprofv3 --kernel-trace --stats -d /tmp/libk -o libk -- python $R/tools/lib_kernel_names.py > /tmp/libk.log 2>&1 < /dev/null
d=$(find /tmp/libk -name '*.db' | head -1)
if [ -z "$d" ]; then echo "no db"; tail -5 /tmp/libk.log; exit 1; fi
python - "$d" <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
for r in cur.execute("select name, count(*), avg(end-start)/1e3 from kernels group by name order by 3 desc"):
    print(f"{r[1]:4d} {r[2]:9.1f} us  {r[0][:600]}")
PY
