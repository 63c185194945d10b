#!/bin/bash
# A/B of two builds of the library on the GPU box: runs the given command with the library as built in the tree and with
# .ab/libplonk_base.so.  usage: tools/ab_run.sh <command...>
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $REPO
echo "== new"; "$@" || exit 1
cp plonk_amd/libplonk_hip.so /tmp/new.so
cp .ab/libplonk_base.so plonk_amd/libplonk_hip.so
echo "== base"; "$@"; rc=$?
cp /tmp/new.so plonk_amd/libplonk_hip.so
echo "== new again"; "$@"
exit $rc
