cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_parity.py tests/test_dp_device.py -m gpu -q -k "full_vocabulary or (follow_the_single and nrms)" 2>&1 | grep -E "^E  |Error|assert " | cut -c1-400 | head -40 > gpurun_out/r2_t16.log
