"""Enum values of include/mlgpu.h, mirrored for Python callers.

tests/test_abi.py parses the header and checks that every value here matches it.
"""

FLOATS_PER_DSPVECTOR = 64  # kFloatsPerDSPVector, reference source/DSP/MLDSPMath.h:8-9


class Status:
    OK = 0
    ERR_INVALID = 1
    ERR_NO_DEVICE = 2
    ERR_HIP = 3
    ERR_OOM = 4
    ERR_UNSUPPORTED = 5
    ERR_RANGE = 6


class Layout:
    QUAD = 0         # [S/4][V][4]  device native
    ROWS = 1         # [T][V][64]   reference DSPVectorArray<V> per vector
    VOICE_MAJOR = 2  # [V][S]
    BROADCAST = 3    # [S] one voice's stream read by every voice (inputs only)


class Op:
    SQRT = 0
    SQRT_APPROX = 1
    ABS = 2
    SIGN = 3
    SIGN_BIT = 4
    SIN = 5
    COS = 6
    LOG = 7
    EXP = 8
    LOG2 = 9
    EXP2 = 10
    SIN_APPROX = 11
    COS_APPROX = 12
    EXP_APPROX = 13
    LOG_APPROX = 14
    LOG2_APPROX = 15
    EXP2_APPROX = 16
    FRACTIONAL_PART = 17
    ROUND_FLOAT_TO_INT = 18
    TRUNCATE_FLOAT_TO_INT = 19
    INT_TO_FLOAT = 20
    UNSIGNED_INT_TO_FLOAT = 21
    EXP_APPROX_OF_SIN_APPROX = 22
    PHASOR_TO_SINE = 23
    ADD = 32
    SUBTRACT = 33
    MULTIPLY = 34
    DIVIDE = 35
    DIVIDE_APPROX = 36
    POW = 37
    POW_APPROX = 38
    MIN = 39
    MAX = 40
    ADD_INT32 = 41
    SUBTRACT_INT32 = 42
    EQUAL = 43
    NOT_EQUAL = 44
    GREATER_THAN = 45
    GREATER_THAN_OR_EQUAL = 46
    LESS_THAN = 47
    LESS_THAN_OR_EQUAL = 48
    PHASOR_TO_SAW = 49
    LERP = 64
    INVERSE_LERP = 65
    CLAMP = 66
    WITHIN = 67
    SELECT = 68
    SELECT_INT = 69
    PHASOR_TO_PULSE = 70

    UNARY = list(range(0, 24))
    BINARY = list(range(32, 50))
    TERNARY = list(range(64, 71))
    # hardware-approximate in the reference (rcpps / rsqrtps): 2^-11 relative tolerance
    HW_APPROX = (1, 36)
    INT_INPUT = (20, 21, 41, 42)


class Vop:
    COLUMN_INDEX = 0
    RANGE_OPEN = 1
    RANGE_CLOSED = 2
    INTERPOLATE_LINEAR = 3
    TABLE = 4           # made by Graph.add(type="const_vector") only


class RowsRule:
    REPEAT = 0
    STRETCH = 1
    SHIFT = 2
    ROTATE = 3
    STRIDED = 4


class Route:
    MULTIPLEX = 0
    MULTIPLEX_LINEAR = 1
    DEMULTIPLEX = 2
    DEMULTIPLEX_LINEAR = 3


class Region:
    UPSAMPLE_2X = 0
    DOWNSAMPLE_2X = 1


class RowOp:
    SUM = 0
    MEAN = 1
    MAX = 2
    MIN = 3


class Proc:
    PHASOR_GEN = 0
    SINE_GEN = 1
    SAW_GEN = 2
    PULSE_GEN = 3
    NOISE_GEN = 4
    TICK_GEN = 5
    IMPULSE_GEN = 6
    ONE_SHOT_GEN = 7
    TEST_SINE_GEN = 8
    LOPASS = 16
    HIPASS = 17
    BANDPASS = 18
    LO_SHELF = 19
    HI_SHELF = 20
    BELL = 21
    ONE_POLE = 32
    DC_BLOCKER = 33
    DIFFERENTIATOR = 34
    INTEGRATOR = 35
    PEAK = 36
    RMS = 37
    ADSR = 38
    GAIN = 48
    INTERPOLATOR1 = 64
    LINEAR_GLIDE = 65
    SAMPLE_ACCURATE_LINEAR_GLIDE = 66
    INTEGER_DELAY = 80
    ALLPASS1 = 81
    FRACTIONAL_DELAY = 82
    PITCHBENDABLE_DELAY = 83
    TEMPO_LOCK = 96
    HALF_BAND = 112            # made by Graph.begin_region / end_region only
    HALF_BAND_BUFFERED = 113

    # processors a bank (chain) can hold; INTERPOLATOR1 / LINEAR_GLIDE are vector-rate: graph nodes only
    ALL = (0, 1, 2, 3, 4, 5, 6, 7, 8, 16, 17, 18, 19, 20, 21, 32, 33, 34, 35, 36, 37, 38, 48, 66, 81)
    VECTOR_RATE = (64, 65, 96)
    DELAYS = (80, 82, 83)      # own per-voice rings in HBM: graph nodes only
    GRAPH_ONLY = (64, 65, 80, 82, 83, 96, 112, 113)
    GENERATORS = (0, 1, 2, 3, 4, 5, 6, 7, 8)
    # outputs pass through sqrtApprox (rsqrtps) in the reference: 2^-11 relative tolerance
    HW_APPROX = (36, 37)
