#!/bin/bash
for f in "" build/variants/*.so; do echo "== ${f:-default}"; if [ -n "$f" ]; then export RT_HIP_LIB=$PWD/$f; else unset RT_HIP_LIB; fi; python scripts/ab_extended.py 16 wf | grep -E "4b|counters" | tail -2; python scripts/bistro_perf.py; done
