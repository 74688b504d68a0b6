#!/bin/bash
# build_ref_variant.sh NAME GITREF [hipcc flags...] -> variants_NAME.so from the library sources of a git revision (A/B against the working tree)
NAME=$1; REF=$2; shift 2
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
T=$(mktemp -d)
mkdir -p $T/continiousenvironment_follower_leader_amd $T/include
git -C "$ROOT" archive "$REF" continiousenvironment_follower_leader_amd/csrc include | tar -x -C $T
cd $T/continiousenvironment_follower_leader_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -pthread "$@" -o "$ROOT/variants_$NAME.so" ftl_abi.hip ftl_scenario.cpp
rm -rf $T
