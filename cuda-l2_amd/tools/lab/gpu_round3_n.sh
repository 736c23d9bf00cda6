#!/bin/bash
# Round-3 last GPU call: the per-geometry PMC table again, for the FINAL tuned table (the late geometries serve >= 5 rows each).
set -u
O=gpurun_out/r3n; mkdir -p $O/pmc_table
export TMPDIR=/tmp
python cuda-l2_amd/tools/pmc_table.py shapes > $O/pmc_shapes.txt
sed -i 's/timeout 240 rocprofv3/timeout 75 rocprofv3/' cuda-l2_amd/tools/pmc_table.sh
bash cuda-l2_amd/tools/pmc_table.sh $O/pmc_table $O/pmc_shapes.txt
python cuda-l2_amd/tools/pmc_table.py table $O/pmc_table $O/pmc_shapes.txt > $O/pmc_table.json 2> $O/pmc_table.err; echo "pmc table rc=$? rows=$(grep -c '"mnk"' $O/pmc_table.json)"
find $O -name "*.db" -delete 2>/dev/null; find $O/pmc_table -name "*kernel_trace.csv" -delete; du -sh $O
