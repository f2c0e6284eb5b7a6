#!/bin/bash
# Stage the reference checkout for tools/runpy_e2e.sh as a git-ignored tarball in the repository root (gpurun carries it to
# the GPU box; it never enters history and is deleted again by the caller after the call).
set -e
cd "$(dirname "$0")/.."
tar czf _ref_stage.tgz -C /root --exclude=__pycache__ --exclude=images --exclude=.git --transform 's/^reference/VideoPose3D/' reference
ls -la _ref_stage.tgz
