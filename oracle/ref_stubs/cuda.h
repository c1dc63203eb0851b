/* empty stand-in: the reference's iou3d_cpu.cpp includes <cuda.h> but its CPU path uses nothing from it (oracle/build_ref.py) */
