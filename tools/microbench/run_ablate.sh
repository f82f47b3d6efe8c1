#!/bin/bash
cd "$(dirname "$0")"
for v in base wt gh wtgh nofft noxchg nobfly empty; do printf "%-8s " $v; ./ablate_$v; done
