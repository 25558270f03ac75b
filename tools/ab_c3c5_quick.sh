#!/bin/bash
# one library: 1M x 1M step (tools/host_turn_clock.py) and configs[4] on one GPU (tools/ab_c5.py, first line)
timeout -k 10 200 python tools/host_turn_clock.py 2>&1 | grep "^C3" ; timeout -k 10 400 python tools/ab_c5.py 2>&1 | grep "^C5" | head -2
