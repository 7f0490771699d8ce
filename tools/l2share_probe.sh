#!/bin/bash
# runs the probe on the product build and the three experiment builds, same box, twice
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
for rep in 1 2; do
  unset DWN_LIB_PATH; python3 tools/l2share_probe.py
  DWN_LIB_PATH=$R/build_ab/xmap/libdwiseneuro_hip.so python3 tools/l2share_probe.py
  DWN_LIB_PATH=$R/build_ab/shy1/libdwiseneuro_hip.so python3 tools/l2share_probe.py shared
  DWN_LIB_PATH=$R/build_ab/shy1x/libdwiseneuro_hip.so python3 tools/l2share_probe.py shared
done
