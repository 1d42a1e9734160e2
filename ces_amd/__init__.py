"""ces_amd -- MI355X-native EKS / ALDI ensemble-update engine.

Drop-in for the ensemble-update hot path of agarbuno/ces
(ces/calibrate.py:241-529).  ``from ces_amd.calibrate import sampling``.
"""
__version__ = "0.1.0"
