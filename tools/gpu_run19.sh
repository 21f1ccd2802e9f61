cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 -L 2>/dev/null | grep -o -E "\b(TCC|MALL|TCP|GL2|EA)[A-Za-z0-9_]*" | sort -u | tr '\n' ' ' > gpurun_out/r2_tcc_counters.txt; wc -w gpurun_out/r2_tcc_counters.txt
rocprofv3 -L 2>/dev/null | grep -i -E "mall|dram|hbm" | head -40 > gpurun_out/r2_mall_lines.txt; cat gpurun_out/r2_mall_lines.txt | cut -c1-200 | head -40
