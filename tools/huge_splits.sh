#!/bin/bash
# the huge geometry with item splits against the library's other choices (tools/time_huge.py; config 3 / config 2 catalogues)
for ns in 0 1 3 5 7; do timeout 100 python tools/time_huge.py c2 50000 lds,huge,huge2 0 $ns 2>&1 | grep "users"; done
for ns in 0 2 4; do timeout 100 python tools/time_huge.py c3 65536 wide,huge 0 $ns 2>&1 | grep "users"; done
for ns in 0 2 3; do timeout 100 python tools/time_huge.py c3 98304 wide,huge 0 $ns 2>&1 | grep "users"; done
for ns in 0 2; do timeout 100 python tools/time_huge.py c3 131072 wide,huge 0 $ns 2>&1 | grep "users"; done
