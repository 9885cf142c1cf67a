#!/bin/bash
# local helper: rebuild the .so (so a stale library never travels), then run a command on the MI355X box
set -e
cd "$(dirname "$0")/../.."
python -m morpheus_amd.build > /dev/null
exec /usr/local/graft/bin/gpurun "$@"
