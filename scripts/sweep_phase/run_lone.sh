#!/bin/bash
cd $(dirname $0)
{
./sweep_probe.bin 0 0 3
./sweep_probe.bin 256 0 3
./sweep_probe.bin 900001 0 3
./sweep_probe.bin 900002 0 3
./sweep_probe.bin 900000 0 3
} 2>&1 | grep -v "^  launch\|^mode"
