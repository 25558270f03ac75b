"""mola-fe-lidar_amd -- MI355X-native ICP registration core for MOLA's LidarOdometry
front-end (the one hot path of MOLAorg/mola-fe-lidar: src/LidarOdometry.cpp:851-895).

csrc/   hand-written HIP kernels (gfx950) + C++ host loop behind the C-ABI of include/mola_icp_amd.h
icp.py  host-side mirror of the reference's ICP surface, marshalling only
synth.py seeded synthetic scan pairs (the reference ships no data)
"""
from . import _lib
from ._lib import (IcpError, NN_AUTO, NN_MFMA, NN_TILED, NN_VALU, TERM_MAX_ITERATIONS, TERM_NO_PAIRINGS, TERM_SOLVER_ERROR,
                   TERM_STALLED, TERM_UNDEFINED)
from .icp import (DevicePool, ICP, Parameters, pool_assignment, Results, pose_from_xyzypr, pose_to_xyzypr, run_loop, run_loop_batch, se3_log,
                  solve_gauss_newton_planes, solve_horn, stall_deltas, mixed_form)

from .lidar_odometry import (CheckResult, LidarOdometry, LidarOdometryParams, Step, check_nonadjacent, montecarlo_guesses,
                             select_checks)

__all__ = ["LidarOdometry", "LidarOdometryParams", "Step", "CheckResult", "check_nonadjacent", "montecarlo_guesses", "select_checks", "ICP", "DevicePool", "pool_assignment", "Parameters", "Results", "IcpError", "pose_from_xyzypr", "pose_to_xyzypr", "se3_log",
           "stall_deltas", "solve_horn", "solve_gauss_newton_planes", "mixed_form", "run_loop", "run_loop_batch", "NN_AUTO", "NN_VALU", "NN_MFMA", "NN_TILED", "TERM_UNDEFINED",
           "TERM_NO_PAIRINGS", "TERM_SOLVER_ERROR", "TERM_MAX_ITERATIONS", "TERM_STALLED"]
