"""Shapes and peaks shared by bench.py and bench_secondary.py (BASELINE.json config 3; MI355X_MICROARCH.md)."""
D_CFG3 = 1024
L_CFG3 = 64
EPS_CFG3 = 0.006
SEED_CFG3 = 20241
C_CFG3 = 65536
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP64_MFMA_PEAK_TFLOPS = 78.6  # dense fp64 matrix-core peak (AMD datasheet; the local guide lists no fp64 figure)
FP64_VECTOR_PEAK_TFLOPS = 78.6  # spec, counts an FMA as 2 flop; kernels here may not contract: ceiling 39.3
