"""Obstacle source of the SE(3) MPC path."""
