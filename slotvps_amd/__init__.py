"""slotvps_amd - MI355X (gfx950) native slot-retriever decode path for Slot-VPS.

Only what the hot path needs: the HIP kernels + C ABI (csrc/, libslotvps_hip.so), their tensor-level
entry points (ops.py) and the host-side mirror of the reference's module / registry interface.
"""
__version__ = "0.1.0"
