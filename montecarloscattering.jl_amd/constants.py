"""cgs constants and compile-time parameters of the path (mirror of include/mcs.h).

Reference: Unitful / UnitfulGaussian / PhysicalConstants.CODATA2018 values used
at src/MonteCarloScattering.jl:10-12, src/constants.jl, src/parameters.jl.
"""
MP = 1.67262192369e-24
ME = 9.1093837015e-28
C = 2.99792458e10
QCGS = 4.803204712570263e-10
KB = 1.380649e-16
SIGMA_T = 6.6524587321e-25
B_CMB0 = 3.27e-6
KEV = 1.602176634e-9          # erg per keV
PSD_MAX = 200                 # parameters.jl:18
NA_C = 100                    # parameters.jl:11
NUM_THERM_BINS = 150          # parameters.jl:20
E_REL_PT = 0.005              # parameters.jl:32
BETA_REL_FL = 0.02            # parameters.jl:30
FLOOR = 1.0e-99
