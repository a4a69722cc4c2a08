for cfg in "8,6,4,3,2,2,2,2,1" "8,8,6,4,3,2,2,2,1" "8,8,8,6,4,3,2,1,1"; do echo "slices $cfg"; bash tools/diag/trace_call.sh tx 4 DBW_SPLIT=1 DBW_SLICES=$cfg | grep "render_fwd\|span"; done
