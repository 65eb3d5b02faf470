# Calibration consumers of the predict path; same module paths as africanus/calibration/.
