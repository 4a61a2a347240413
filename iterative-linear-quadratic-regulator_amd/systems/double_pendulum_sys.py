"""Fully actuated double pendulum, n_x = 4, n_u = 2 (torques on both joints).

Reference: python/class_files/systems/double_pendulum_sys.py:9-206 (identical physics
to the under-actuated variant, f_act = [tau1, tau2] at :202).
Device code: csrc/dynamics.hpp ``DoublePendulum<T, 2>``.
"""
from .. import _lib
from .UA_double_pendulum_sys import MyUADoublePendulum


class MyDoublePendulum(MyUADoublePendulum):
    SYSTEM_ID = _lib.SYS_DOUBLE_PENDULUM
    N_U = 2
