#!/bin/bash
# builds tools/standin/libtransport_standin.so (gfx950); the .so is git-ignored and travels to the GPU box with the snapshot
cd "$(dirname "$0")" && hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 transport_standin.hip -o libtransport_standin.so
