# Convolutional degridding; same module paths as africanus/gridding/.
