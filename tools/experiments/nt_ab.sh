#!/bin/bash
# Runs on the GPU box (round 5): cache policy of the sweep scratch — non-temporal loads in the second sweep (nt1), loads and
# stores (nt2) — against the shipped library.  Output gpurun_out/ab_nt.txt
exec tools/experiments/ab.sh nt base nt1 nt2
