#!/bin/bash
# convergence statistics of the five configs (debug aid): tools/robust_sweep.sh [max_iter]
MI=${1:-1000}
python tools/solve_stats.py pendulum 50 256 $MI 2>&1 | grep '^{' | cut -c1-330
python tools/solve_stats.py car 500 128 $MI 2>&1 | grep '^{' | cut -c1-330
python tools/solve_stats.py acrobot 101 256 $MI 2>&1 | grep '^{' | cut -c1-330
python tools/solve_stats.py cartpole 200 128 $MI 2>&1 | grep '^{' | cut -c1-330
python tools/solve_stats.py acrobot 1000 128 $MI 2>&1 | grep '^{' | cut -c1-330
