set -x
mkdir -p gpurun_out
AMD_SERIALIZE_KERNEL=3 python tools/dbg_decode.py parseq; AMD_SERIALIZE_KERNEL=3 python tools/dbg_decode.py parseq-tiny > gpurun_out/r2_dbg_decode.log 2>&1; tail -15 gpurun_out/r2_dbg_decode.log
python tools/x3_diag.py > gpurun_out/r2_x3_diag.log 2>&1; tail -20 gpurun_out/r2_x3_diag.log
