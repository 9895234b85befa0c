#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/plane_streams.hip -o /tmp/plane_streams && /tmp/plane_streams
