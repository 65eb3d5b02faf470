"""The workload table of bench.py (--workload)."""
from .workloads_dft import Dft, DftF32, GaussDft
from .workloads_fused import FusedDde, FusedDdeAnt128, FusedDdeAntC64, FusedDdeC64
from .workloads_grid import Degrid, Wgrid, WgridF32Planes

WORKLOADS = {"dft": Dft, "dft_complex": Dft, "dft_f32": DftF32, "gauss": GaussDft, "fused_dde": FusedDde,
             "fused_dde_ant": FusedDde, "fused_dde_ant128": FusedDdeAnt128, "fused_dde_ant_c64": FusedDdeAntC64, "fused_dde_c64": FusedDdeC64, "degrid": Degrid, "wgrid": Wgrid, "wgrid_f32planes": WgridF32Planes}
METRIC = "Mvis/s (rows x chans) for predict_vis at 1e6 rows/64 ch/1000 src; fp64 max-abs err"
