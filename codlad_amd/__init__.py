"""MI355X-native drop-in for CODLAD's sampling path (HIP kernels behind include/codlad_hip.h)."""
import os

# The HIP runtime keeps kernel arguments in host memory unless told otherwise; the first scalar load of every wave of
# every launch then crosses PCIe.  A DDPM step of a small job is 17 short dependent launches: one 87-residue protein
# takes 253 us per step with host-side arguments and 215 us with device-side ones (tools/small_job_latency.py), large
# jobs gain ~0.5 %.  The runtime reads the variable when it initialises, i.e. at the process's first HIP call, which
# comes after this import in every entry point of the package; a value set by the caller wins.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
