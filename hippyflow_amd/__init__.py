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
from .multivector import (MatMvMult, MatMvTranspmult, MultiVector, MvDSmatMult, Vector, dense_to_mv_local, ingest_stream, mv_to_dense,
                          mv_to_dense_local)
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
from .io_utils import get_projectors, modify_projectors, spectrum_plot
from .datasets import (derivative_dataset, jacobian_svds, jacobian_times_input_basis,
                       jacobian_transpose_times_output_basis, save_JPsi_data, save_JstarPhi_data, save_Jsvd_data,
                       save_mq_data)
from .datagen import DataGenerator, compress_dataset, data_generator_settings

__version__ = "0.1.0"

# ---- the FEniCS / hIPPYlib side of the reference package
# `from hippyflow import *` also hands a driver the classes that BUILD the host objects (confusion_problem_setup.py:94
# `BiLaplacian2D(...)`, confusion_linear_observable.py:148 `LinearStateObservable(pde, B)`).  They are PDE-side code
# (dolfin / hippylib; SURVEY section 2 "out of scope"), so nothing of them is re-implemented here: when the reference
# package is installed next to FEniCS they are FORWARDED, name by name and on first use, so that a driver's
# `from hippyflow import *` line can become `from hippyflow_amd import *` with nothing else changed.  Without the
# reference package the names do not exist (AttributeError that says so); nothing on the device path needs them.
_HOST_SIDE_NAMES = {
    "BlockVector": "hippyflow.modeling.blockVector",
    "ConstrainedNSolver": "hippyflow.modeling.cMinimization",
    "hippylibModelWrapper": "hippyflow.modeling.hippylibModelWrapper",
    "hippylibModelWrapperSettings": "hippyflow.modeling.hippylibModelWrapper",
    "BiLaplacian2D": "hippyflow.modeling.maternPrior",
    "Laplacian2D": "hippyflow.modeling.maternPrior",
    "MultiPDEProblem": "hippyflow.modeling.multiPDEProblem",
    "LinearStateObservable": "hippyflow.modeling.observable",
    "DomainRestrictedOperator": "hippyflow.modeling.observable",
    "hippylibModelLinearStateObservable": "hippyflow.modeling.observable",
    "read_serial_write_parallel_mesh": "hippyflow.utilities.mesh_utils",
    # plotting helpers of `from hippyflow import *` (spectrum_plot has its own, matplotlib-only counterpart in io_utils)
    "generic_semilogy_plot": "hippyflow.utilities.plotting",
    "plot_accs_vs_data": "hippyflow.utilities.plotting",
    "plot_singular_values_with_std": "hippyflow.utilities.plotting",
    "subspace_angle_video": "hippyflow.utilities.plotting",
    "plot_eigenvector": "hippyflow.utilities.plot_eigenvectors",
}


def _reference_package_importable():
    import importlib.util
    try:
        return all(importlib.util.find_spec(m) is not None for m in ("hippyflow", "dolfin", "hippylib"))
    except (ImportError, ValueError):
        return False


def __getattr__(name):
    where = _HOST_SIDE_NAMES.get(name)
    if where is None:
        raise AttributeError("module 'hippyflow_amd' has no attribute %r" % name)
    import importlib
    try:
        value = getattr(importlib.import_module(where), name)
    except Exception as exc:       # ImportError of dolfin / hippylib, or a reference version without the name
        raise AttributeError("hippyflow_amd.%s is the reference package's own %s.%s (FEniCS / hIPPYlib side, forwarded, "
                             "not re-implemented): %s: %s" % (name, where, name, type(exc).__name__, exc)) from exc
    globals()[name] = value
    return value


# star-import: everything public above, plus the forwarded names when the reference package can provide them
__all__ = [_n for _n, _v in list(globals().items()) if not _n.startswith("_") and type(_v).__name__ != "module"]
if _reference_package_importable():
    for _n in sorted(_HOST_SIDE_NAMES):      # only what this installation of the reference really has (star-import needs every name)
        try:
            __getattr__(_n)
            __all__.append(_n)
        except AttributeError:
            pass
