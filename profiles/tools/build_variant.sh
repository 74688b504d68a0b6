#!/bin/bash
# build_variant.sh NAME [hipcc flags...] -> variants_NAME.so at the repo root (git-ignored; travels to the GPU box): A/B builds of the library
NAME=$1; shift
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$ROOT/continiousenvironment_follower_leader_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -pthread "$@" -o "$ROOT/variants_$NAME.so" ftl_abi.hip ftl_scenario.cpp
