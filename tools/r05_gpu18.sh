#!/bin/bash
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/addtid_probe.hip -o /tmp/addtid_probe 2>&1 | grep -v warning | head -5
timeout 60 /tmp/addtid_probe
