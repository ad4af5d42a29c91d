#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the order-defining host logic (quadtree, grid hand-off, stereo
# post-filter, YOLO NMS) at edge sizes.  CPU only.
set -e
cd "$(dirname "$0")/../.."
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
$CXX -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -Idani_slam_amd/csrc -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ \
    tools/sanitize/san_host_logic.cpp dani_slam_amd/csrc/host_logic.cpp -o /tmp/san_host_logic
/tmp/san_host_logic
