# Same module layout as africanus/gridding/perleypolyhedron (degridder + kernels).
