"""hippyflow_amd -- MI355X-native (gfx950) randomized double-pass eigensolve behind hippyflow's
model-based projectors (ActiveSubspaceProjector, KLEProjector, PODProjector).

Host code is Python over a C ABI (include/hfmi.h, hippyflow_amd/libhfmi.so: hand-written HIP).
There is no CPU fallback: importing the package is cheap and GPU-free, but any compute call
raises if libhfmi.so or a GPU is missing.
"""
from ._lib import Context, HfmiError, build_tag, device_count, load, pinned_empty
from .collectives import (CollectiveOperator, MatrixMultCollectiveOperator, MultipleSamePartitioningPDEsCollective,
                          MultipleSerialPDEsCollective, NativeCollective, NullCollective, TorchCollective,
                          checkFunctionSpaceConsistentPartitioning, checkMeshConsistentPartitioning, splitCommunicators)
from .hostvec import ADJOINT, CONTROL, PARAMETER, STATE, HostMultiVector, HostVector, new_host_vector, set_host_vector_factory
from .multivector import MatMvMult, MatMvTranspmult, MultiVector, MvDSmatMult, Vector, ingest_stream
from .operators import (ComposedOperator, CsrOperator, CsrPCGSolver, DenseJacobianOperator, DeviceOperator, HostCallbackOperator,
                        LowRankOperator, LowRankRectangularOperator, MassPreconditionedCovarianceOperator,
                        JJT, JTJ, Jacobian, MeanJJTfromDataOperator, MeanJTJfromDataOperator, ObservableControlJacobian,
                        ObservableJacobian, PriorPreconditionedProjector,
                        SeriallySampledJacobianOperator, StateSpaceIdentityOperator, npToDolfinOperator,
                        SnapshotGramOperator, Solver2Operator, SummedListOperator, as_device_operator, npToDeviceOperator)
from .projectors import (ActiveSubspaceParameterList, ActiveSubspaceProjector, BoundaryRestrictedKLEProjector,
                         KLEParameterList, KLEProjector,
                         ParameterList, PODParameterList, PODProjector, PODProjectorFromData, weighted_l2_norm_vector)
from .randomized import accuracyEnhancedSVD, doublePass, doublePassG, parRandom, svd_small, sym_eig_small
from .errors import input_output_error_test, projection_error_test
from .io_utils import get_projectors, modify_projectors
from .datasets import (derivative_dataset, jacobian_svds, jacobian_times_input_basis,
                       jacobian_transpose_times_output_basis, save_JPsi_data, save_JstarPhi_data, save_Jsvd_data,
                       save_mq_data)
from .datagen import DataGenerator, compress_dataset, data_generator_settings
from .utilities import dense_to_mv_local, mv_to_dense, mv_to_dense_local

__version__ = "0.1.0"
