cp das3r_amd/libdas3r_hip.so /tmp/libdas3r_hip.default.so
for v in default u4 u1 b256 default; do
  if [ "$v" = default ]; then cp /tmp/libdas3r_hip.default.so das3r_amd/libdas3r_hip.so; else cp das3r_amd/libdas3r_hip.$v.so das3r_amd/libdas3r_hip.so; fi
  echo "#### variant $v"
  python tools/probes/job_phases.py 2>&1 | grep -E "deg0|deg1|test_pass|PSNR"
done
cp /tmp/libdas3r_hip.default.so das3r_amd/libdas3r_hip.so
