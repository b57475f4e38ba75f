"""Model-based projectors: the callers of the randomized eigensolve (SURVEY.md section 8a a10-a12, 8b
"Projector classes").  Same class names, constructor arguments, parameter keys/defaults, method
names, return tuples, stored attributes and saved-file names as the reference:

  ActiveSubspaceProjector / ActiveSubspaceParameterList   hippyflow/modeling/activeSubspaceProjector.py:33-66,252-673
  KLEProjector / KLEParameterList                         hippyflow/modeling/KLEProjector.py:30-45,72-199
  PODProjector / PODParameterList                         hippyflow/modeling/PODProjector.py:35-49,52-389
  PODProjectorFromData                                    hippyflow/modeling/PODProjector.py:666-852

FEniCS/hIPPYlib PDE work stays a host black box, reached through the REFERENCE'S OWN protocols: an ``observable``
with ``generate_vector / init_vector / solveFwd / evalu / setLinearizationPoint / applyC / solveFwdIncremental / applyB`` and
their adjoint counterparts (modeling/observable.py:66-323), a ``prior`` with ``init_vector(x, "noise")``,
``sample(noise, m)``, ``mean``, ``R`` / ``Rsolver`` (or ``Hlr``), ``M`` / ``Msolver`` -- all on dolfin-like host vectors
(``hostvec.new_host_vector``).  The sampling loops of the reference run as they are written there
(PODProjector.py:343-357, activeSubspaceProjector.py:163-248,347-397); what they produce is streamed into HBM through
pinned buffers (``ingest_stream`` / ``upload_async``) and contracted on the device.

Observables that already hold their data on the device may short-cut the loops with optional hooks:
``sample_observables(n, prior, noise)`` / ``observable_stream(n, prior, noise)`` (snapshots), ``jacobian_data(n)`` /
``jacobian_stream(n)`` (stored Jacobians), ``jtj_host_operator()`` / ``jjt_host_operator()`` (a ready host operator); a
prior may offer ``sample_block(n)`` and an explicit covariance ``C``.
"""
import os
import time

import numpy as np

from . import _lib as L
from .collectives import CollectiveOperator, MatrixMultCollectiveOperator, NullCollective
from . import hostvec as H
from .multivector import ingest_stream, MatMvMult, MultiVector, mv_to_dense
from .operators import (CsrOperator, CsrPCGSolver, DeviceOperator, HostCallbackOperator, MassPreconditionedCovarianceOperator,
                        MeanJJTfromDataOperator, MeanJTJfromDataOperator, ObservableJacobian, SeriallySampledJacobianOperator,
                        SnapshotGramOperator, Solver2Operator, as_device_operator)
from .randomized import doublePass, doublePassG, parRandom, sym_eig_small


class ParameterList(object):
    """hp.ParameterList: {name: [default, help]} with item access on the values."""

    def __init__(self, data):
        self.data = data

    def __getitem__(self, key):
        if key in self.data:
            return self.data[key][0]
        raise ValueError(key)

    def __setitem__(self, key, value):
        if key in self.data:
            self.data[key][0] = value
        else:
            raise ValueError(key)

    def __contains__(self, key):
        return key in self.data

    def keys(self):
        return self.data.keys()

    def showMe(self, indent=""):
        for k in sorted(self.data.keys()):
            print(indent, "---")
            print(indent, k, "(default):", self.data[k][0])
            print(indent, self.data[k][1])


def ActiveSubspaceParameterList():
    """activeSubspaceProjector.py:33-66 (same keys, same defaults)."""
    parameters = {}
    parameters['samples_per_process'] = [64, 'Monte-Carlo Jacobian samples averaged on each rank (one GPU per rank)']
    parameters['jacobian_data_per_process'] = [512, 'per-rank samples whose low-rank Jacobians are written out as training data']
    parameters['error_test_samples'] = [50, 'held-out draws used by the projection-error tests']
    parameters['rank'] = [128, 'r: eigenpairs kept (columns of the decoder)']
    parameters['jacobian_rank'] = [128, 'truncation rank of the per-sample Jacobian SVDs']
    parameters['control_jacobian_rank'] = [None, 'the same for Jacobians with respect to the control variable (None: not formed)']
    parameters['oversampling'] = [10, 'p: extra probe vectors, the double pass works on r + p']
    parameters['double_loop_samples'] = [20, 'inner samples of the nested Monte-Carlo estimator']
    parameters['verbose'] = [True, 'print progress and timings on rank 0']
    parameters['input_decoder_name'] = ['_input_decoder', 'tail of the input-decoder file name (after AS_<nsamples>)']
    parameters['output_decoder_name'] = ['_output_decoder', 'tail of the output-decoder file name']
    parameters['initialize_samples'] = [False, 'draw and store the linearisation points when the projector is built']
    parameters['serialized_sampling'] = [True, 'one sample at a time through the host operator instead of stored Jacobians']
    parameters['observable_constructor'] = [None, 'callable(mesh, **kwargs) building one more local observable']
    parameters['observable_kwargs'] = [{}, 'keyword arguments handed to observable_constructor']
    parameters['output_directory'] = [None, 'prefix of every file written (decoders, eigenvalues, plots)']
    parameters['plot_label_suffix'] = ['', 'appended to plot titles and file names']
    parameters['save_and_plot'] = [True, 'write the arrays and spectra (the unit tests switch it off)']
    parameters['store_Omega'] = [False, 'keep the probe block on the projector so that a test can replay it']
    parameters['ms_given'] = [False, 'linearisation points are supplied by the caller (tests of the serialized route)']
    return ParameterList(parameters)


def PODParameterList():
    """PODProjector.py:35-49."""
    parameters = {}
    parameters['sample_per_process'] = [100, 'state snapshots drawn on each rank for the POD']
    parameters['data_per_process'] = [250, 'per-rank (parameter, observable) pairs written out as training data']
    parameters['rank'] = [20, 'r: POD modes kept']
    parameters['oversampling'] = [10, 'p: extra probe vectors, the double pass works on r + p']
    parameters['verbose'] = [True, 'print progress and timings on rank 0']
    parameters['output_directory'] = [None, 'prefix of every file written']
    parameters['plot_label_suffix'] = ['', 'appended to plot titles and file names']
    return ParameterList(parameters)


def KLEParameterList():
    """KLEProjector.py:30-45."""
    parameters = {}
    parameters['error_test_samples'] = [50, 'held-out prior draws used by the projection-error test']
    parameters['rank'] = [128, 'r: Karhunen-Loeve modes kept']
    parameters['oversampling'] = [10, 'p: extra probe vectors, the double pass works on r + p']
    parameters['verbose'] = [True, 'print progress and timings on rank 0']
    parameters['output_directory'] = ['./data/', 'prefix of every file written']
    parameters['plot_label_suffix'] = ['', 'appended to plot titles and file names']
    parameters['save_and_plot'] = [True, 'write the decoder, the eigenvalues and the spectrum plot']
    parameters['input_decoder_name'] = ['KLE_decoder', 'file name of the decoder array (without .npy)']
    return ParameterList(parameters)


def _is_root(collective=None):
    """Rank 0 of the sample-parallel communicator writes files and prints (the reference tests
    ``mesh_constructor_comm.rank == 0`` / world rank 0, activeSubspaceProjector.py:470-480)."""
    if collective is not None and hasattr(collective, "rank"):
        return int(collective.rank()) == 0
    return int(os.environ.get("RANK", "0")) == 0


def _save(directory, name, array):
    """np.save(output_directory + name), the reference's convention (activeSubspaceProjector.py:475-480,
    PODProjector.py:382-384, KLEProjector.py:190-192): ``output_directory`` is a PREFIX ('out/' or 'out/run1_')."""
    parent = os.path.dirname(directory + name)
    if parent:
        os.makedirs(parent, exist_ok=True)
    np.save(directory + name, array)


def _plot_spectrum(directory, name, d, title):
    """``<output_directory><name>.pdf`` next to the saved arrays, as the reference leaves it (activeSubspaceProjector.py:482-485,
    KLEProjector.py:194-197, PODProjector.py:386-389); a no-op without matplotlib."""
    from .io_utils import spectrum_plot
    try:
        spectrum_plot(d, axis_label=['i', r'$\lambda_i$', title], out_name=directory + name + '.pdf')
    except Exception:          # noqa: BLE001 -- a plot must never fail a solve (fonts, mathtext, read-only directories ...)
        pass


def _plot_singular_values(path, sigmas):
    from .io_utils import singular_values_plot
    try:
        sigmas = np.asarray(sigmas)
        singular_values_plot(np.mean(sigmas, axis=0), np.std(sigmas, axis=0), outname=path)
    except Exception:          # noqa: BLE001 -- cosmetics never fail a run
        pass


def _draw_omega(N, nvec, collective, ctx, stored=None):
    """Probe block.  The reference draws on rank 0 and broadcasts k vectors of length N
    (activeSubspaceProjector.py:433-443,536-551).  Here rank 0's SHARED generator state (seed, shared stream) -- 16 bytes
    -- is what is broadcast; every rank then regenerates the same counter-based Philox block in its own HBM and EVERY
    rank moves on to the next shared stream.  Shared draws live under their own Philox key (namespace 0,
    ``randomized._ParRandom``), disjoint from the per-rank private streams the sampling loops draw their noise from, so a
    probe can never coincide with a Monte-Carlo sample on any rank.  A stored Omega (unit-test path) is broadcast from
    rank 0 as a block."""
    if stored is not None:
        Omega = MultiVector(stored)
        collective.bcast(Omega, root=0)
        return Omega
    state = np.array([parRandom.seed & 0xFFFFFFFF, parRandom.shared_stream & 0xFFFFFFFF], dtype=np.uint64)
    state = collective.bcast(state, root=0)
    draw = type(parRandom)(int(state[0]))
    draw.shared_stream = int(state[1])
    Omega = MultiVector(int(N), int(nvec), ctx=ctx)
    draw.normal(1., Omega, shared=True)
    parRandom.shared_stream = int(state[1]) + 1
    return Omega


def _speaks_reference_protocol(observable):
    """An observable of the reference (modeling/observable.py): PDE solves behind solveFwd / setLinearizationPoint."""
    return hasattr(observable, "solveFwd") and hasattr(observable, "generate_vector")


def _operator_size(op, dim=0):
    """Length of the vectors an operator acts on: ``shape`` of a device operator, else whatever its ``init_vector`` makes."""
    if hasattr(op, "shape"):
        return int(op.shape[0])
    return int(H.shape_with(op.init_vector, dim).size())


def _prior_draws(prior, noise, m, n):
    """n prior samples as numpy rows, by the reference's loop: hp.parRandom.normal(1, noise); prior.sample(noise, m)."""
    for _ in range(n):
        parRandom.normal(1, noise)
        prior.sample(noise, m)
        yield m.get_local()


def _prior_sample_block(prior, n, noise=None, observable=None):
    """n prior draws for the projection-error tests: ``prior.sample_block(n)`` when the prior offers it, else the
    reference's loop (KLEProjector.py:222-231, activeSubspaceProjector.py:1075-1090) over host vectors."""
    if hasattr(prior, "sample_block"):
        return prior.sample_block(n)
    if noise is None:
        noise = H.new_host_vector()
        prior.init_vector(noise, "noise")
    if observable is not None and hasattr(observable, "generate_vector"):
        m = observable.generate_vector(H.PARAMETER)
    else:
        m = H.shape_with(prior.init_vector, 0)
    return np.stack(list(_prior_draws(prior, noise, m, n)))


def _observable_draws(observable, prior, control_distribution, n, noise):
    """The reference's snapshot loop (PODProjector.py:343-357), one observable per item: noise, prior.sample, control,
    forward solve, ``evalu``."""
    u = observable.generate_vector(H.STATE)
    m = observable.generate_vector(H.PARAMETER)
    z = None if control_distribution is None else observable.generate_vector(H.CONTROL)
    for _ in range(n):
        parRandom.normal(1, noise)
        prior.sample(noise, m)
        point = [u, m, None]
        if control_distribution is not None:
            z.zero()
            control_distribution.sample(z)
            point.append(z)
        observable.solveFwd(u, point)
        yield observable.evalu(u).get_local()


def _observable_samples(observable, prior, control_distribution, n, noise, ctx):
    """n observable samples in HBM (one per vector): the observable's own bulk hooks if it has them, else the
    reference's loop with each result on its way to the device while the next PDE is being solved."""
    if hasattr(observable, 'observable_stream'):
        return ingest_stream(observable.observable_stream(n, prior, None), n, 1, observable.output_dimension(), ctx=ctx)
    if hasattr(observable, 'sample_observables'):
        X = observable.sample_observables(n, prior, None)
        return X if isinstance(X, MultiVector) else MultiVector.from_vectors(X, ctx=ctx)
    if noise is None:
        noise = H.new_host_vector(observable.mpi_comm() if hasattr(observable, "mpi_comm") else None)
        prior.init_vector(noise, "noise")
    q = H.shape_with(observable.init_vector, 0).size()
    return ingest_stream(_observable_draws(observable, prior, control_distribution, n, noise), n, 1, q, ctx=ctx)


# =====================================================================================
# Active subspace
# =====================================================================================
class ActiveSubspaceProjector:
    """Projectors from the sample-averaged GN Hessian E[J^T J] (input) and E[J J^T] (output)."""

    def __init__(self, observable, prior, control_distribution=None, mesh_constructor_comm=None,
                 collective=None, parameters=None, ctx=None):
        # defaults made per instance: the reference's `parameters = ActiveSubspaceParameterList()` default is ONE list shared by every
        # projector built without one (a rank raised by test_errors on one instance would leak into the next)
        self.parameters = parameters if parameters is not None else ActiveSubspaceParameterList()
        self.observable = observable
        self.prior = prior
        self.control_distribution = control_distribution
        if mesh_constructor_comm is None and hasattr(observable, "mpi_comm"):
            mesh_constructor_comm = observable.mpi_comm()                                   # :279-282
        self.mesh_constructor_comm = mesh_constructor_comm
        self.collective = collective if collective is not None else NullCollective()
        self.ctx = ctx or L.Context.default()
        self.noise = None
        if _speaks_reference_protocol(observable) and hasattr(prior, "init_vector"):
            self.noise = H.new_host_vector(mesh_constructor_comm)                           # :309-310
            prior.init_vector(self.noise, "noise")
        self.ms = None            # linearisation points: given by the caller (ms_given) or kept from the batched draw
        self.zs = None
        self.qs = None            # observables at the stored samples (batched route)
        self._zs_kept = None
        self.Js = None            # (block, ndata, q) once materialised
        self.d_GN = None
        self.V_GN = None
        self.d_GN_noprior = None
        self.V_GN_noprior = None
        self.prior_preconditioned = None
        self.d_NG = None
        self.U_NG = None
        self.Omega_GN = None
        self.Omega_NG = None
        if self.parameters['initialize_samples'] and not self.parameters['serialized_sampling']:
            self._initialize_batched_samples()                                              # :330-332

    # ---- data source -----------------------------------------------------------------
    def _reference_jacobian_rows(self, n):
        """The reference's batched initialisation (:347-397) with HBM as the store: per sample -- noise, prior.sample,
        control, forward solve (a failing solve is answered with a fresh draw, as upstream), linearisation -- and then
        the q rows of that sample's Jacobian (one adjoint solve each) as one item for the ingest stream.  One observable
        serves every sample: what the reference keeps alive as ``samples_per_process`` FEniCS copies
        (``observable_constructor``) is here a (q x N) slab per sample in device memory."""
        sampler = SeriallySampledJacobianOperator(self.observable, self.noise, self.prior,
                                                  control_distribution=self.control_distribution, operation='JTJ', nsamples=n,
                                                  ms=self.ms if self.parameters['ms_given'] else None, zs=self.zs,
                                                  jacobian_factory=ObservableJacobian)
        J = None
        kept, qs, zs = [], [], []
        for _ in sampler._points():
            J = J or ObservableJacobian(self.observable)
            if not self.parameters['ms_given']:
                kept.append(sampler.m.get_local())
            qs.append(self.observable.evalu(sampler.u).get_local())       # q_i = B u_i of the stored sample (:989)
            if self.control_distribution is not None and getattr(sampler, 'z', None) is not None:
                zs.append(sampler.z.get_local())
            yield J.rows()
        if kept:
            self.ms = kept
        self.qs = qs
        self._zs_kept = zs or None

    def _initialize_batched_samples(self):
        """Materialise this rank's Jacobians in HBM (counterpart of :347-397)."""
        n = self.parameters['samples_per_process']
        if hasattr(self.observable, 'jacobian_stream'):
            # one Jacobian per host PDE solve (:178-221): sample i + 1 is produced while sample i is on its way to HBM
            # through a pinned buffer (multivector.ingest_stream)
            q, dM = self.observable.jacobian_shape()
            block = ingest_stream(self.observable.jacobian_stream(n), n, q, dM, ctx=self.ctx)
            self.Js = (block, int(n), int(q))
            return
        if not hasattr(self.observable, 'jacobian_data') and _speaks_reference_protocol(self.observable):
            if self.parameters['ms_given']:
                assert self.ms is not None
                n = len(self.ms)
            q, dM = ObservableJacobian(self.observable).shape
            block = ingest_stream(self._reference_jacobian_rows(n), n, q, dM, ctx=self.ctx)
            self.Js = (block, int(n), int(q))
            return
        data = self.observable.jacobian_data(n)
        if isinstance(data, tuple):
            block, ndata, q = data
        else:
            data = np.asarray(data, dtype=np.float64)
            ndata, q, dM = data.shape
            block = MultiVector.from_vectors(data.reshape(ndata * q, dM), ctx=self.ctx)
        assert ndata == n, "observable returned %d samples, samples_per_process is %d" % (ndata, n)
        self.Js = (block, int(ndata), int(q))

    def _linearize_at_mean_if_needed(self):
        """:523-541 -- the incremental blocks of the PDE problem exist only after a first linearisation."""
        problem = getattr(self.observable, "problem", None)
        if problem is None or getattr(problem, "C", True) is not None:
            return
        m_mean = self.prior.mean
        if hasattr(problem, 'parameter_projection'):
            m_mean = problem.parameter_projection(m_mean)
        point = [problem.generate_state(), m_mean, None]
        if self.control_distribution is not None:
            if hasattr(self.control_distribution, 'mean'):
                point.append(self.control_distribution.mean)
            else:
                z = self.observable.generate_vector(H.CONTROL)
                self.control_distribution.sample(z)
                point.append(z)
        problem.solveFwd(point[0], point)
        self.observable.setLinearizationPoint(point)

    def _local_operator(self, operation):
        """The per-rank averaged operator.  Batched: device kernels over stored Jacobians.  Serialized: the reference's
        ``SeriallySampledJacobianOperator`` over the observable's own PDE calls, re-sampled on every application."""
        if self.parameters['serialized_sampling']:
            if not self.parameters['ms_given'] and hasattr(self.observable, 'jtj_host_operator'):
                host = self.observable.jtj_host_operator() if operation == 'JTJ' else self.observable.jjt_host_operator()
                n = self.observable.input_dimension() if operation == 'JTJ' else self.observable.output_dimension()
                return HostCallbackOperator(host, n, ctx=self.ctx)
            if _speaks_reference_protocol(self.observable) and not hasattr(self.observable, 'jacobian_data'):
                if self.parameters['ms_given']:                                           # :504-511
                    assert self.ms is not None
                    if self.control_distribution is not None:
                        assert self.zs is not None and self.zs[0] is not None
                    op = SeriallySampledJacobianOperator(self.observable, self.noise, self.prior, operation=operation,
                                                         ms=self.ms, zs=self.zs)
                else:
                    op = SeriallySampledJacobianOperator(self.observable, self.noise, self.prior,
                                                         control_distribution=self.control_distribution, operation=operation,
                                                         nsamples=self.parameters['samples_per_process'])
                self._linearize_at_mean_if_needed()
                return op
        if self.Js is None:
            self._initialize_batched_samples()
        block, ndata, q = self.Js
        if operation == 'JTJ':
            return MeanJTJfromDataOperator.from_block(block, ndata, q, prior=self.prior)
        return MeanJJTfromDataOperator((block, ndata, q))

    # ---- public API ------------------------------------------------------------------
    def construct_input_subspace(self, prior_preconditioned=True, name_suffix=None):
        if self.parameters['serialized_sampling']:
            return self._construct_serialized_jacobian_subspace(prior_preconditioned=prior_preconditioned, operation='JTJ',
                                                                name_suffix=name_suffix)
        return self._construct_input_subspace_batched(prior_preconditioned=prior_preconditioned, name_suffix=name_suffix)

    def construct_output_subspace(self, name_suffix=None):
        if self.parameters['serialized_sampling']:
            return self._construct_serialized_jacobian_subspace(operation='JJT', name_suffix=name_suffix)
        return self._construct_output_subspace_batched(name_suffix=name_suffix)

    def _construct_input_subspace_batched(self, prior_preconditioned=True, name_suffix=None):
        return self._construct(operation='JTJ', prior_preconditioned=prior_preconditioned, name_suffix=name_suffix,
                               wrapper=CollectiveOperator)

    def _construct_output_subspace_batched(self, name_suffix=None):
        return self._construct(operation='JJT', prior_preconditioned=False, name_suffix=name_suffix, wrapper=CollectiveOperator)

    def _construct_serialized_jacobian_subspace(self, prior_preconditioned=True, operation='JTJ', name_suffix=None):
        return self._construct(operation=operation, prior_preconditioned=prior_preconditioned and operation == 'JTJ',
                               name_suffix=name_suffix, wrapper=MatrixMultCollectiveOperator)

    def _construct(self, operation, prior_preconditioned, name_suffix, wrapper):
        t0 = time.time()
        local_op = self._local_operator(operation)
        # This averaging assumes every process has an equal number of samples (reference :429-430,509-510)
        average_op = wrapper(local_op, self.collective, mpi_op='avg')
        N = _operator_size(local_op)
        nvec = self.parameters['rank'] + self.parameters['oversampling']
        stored = self.Omega_GN if operation == 'JTJ' else self.Omega_NG
        Omega = _draw_omega(N, nvec, self.collective, self.ctx, stored=stored)
        if self.parameters['store_Omega'] and stored is None:
            if operation == 'JTJ':
                self.Omega_GN = Omega
            else:
                self.Omega_NG = Omega

        if operation == 'JTJ':
            if prior_preconditioned:
                if hasattr(self.prior, "R"):
                    B, Binv = self.prior.R, self.prior.Rsolver
                else:
                    B, Binv = self.prior.Hlr, self.prior.Hlr
                self.d_GN, self.V_GN = doublePassG(average_op, B, Binv, Omega, self.parameters['rank'], s=1)
                as_decoder = self.V_GN
                as_encoder = MultiVector(as_decoder)
                MatMvMult(as_device_operator(B, N, self.ctx), as_decoder, as_encoder)
            else:
                self.d_GN, self.V_GN = doublePass(average_op, Omega, self.parameters['rank'], s=1)
                as_decoder = self.V_GN
                as_encoder = MultiVector(as_decoder)
            self.prior_preconditioned = prior_preconditioned
            self._input_subspace_construction_time = time.time() - t0
            result = (self.d_GN, as_decoder, as_encoder)
        else:
            self.d_NG, self.U_NG = doublePass(average_op, Omega, self.parameters['rank'], s=1)
            output_decoder = self.U_NG
            output_encoder = MultiVector(output_decoder)
            self._output_subspace_construction_time = time.time() - t0
            result = (self.d_NG, output_decoder, output_encoder)

        if self.parameters['verbose'] and _is_root(self.collective):
            which = 'Input' if operation == 'JTJ' else 'Output'
            print((which + ' subspace construction took ' + str(time.time() - t0)[:5] + ' s').center(80))
        if self.parameters['save_and_plot'] and _is_root(self.collective) and self.parameters['output_directory'] is not None:
            name = 'AS_' + str(int(self.parameters['samples_per_process'] * self.collective.size()))
            if name_suffix is not None:
                assert type(name_suffix) is str
                name += name_suffix
            out = self.parameters['output_directory']
            suffix, r_str = self.parameters['plot_label_suffix'], str(self.parameters['rank'])
            if operation == 'JTJ':
                _save(out, name + self.parameters['input_decoder_name'], mv_to_dense(self.V_GN))
                _save(out, name + '_d_GN', self.d_GN)
                _plot_spectrum(out, name + '_input_eigenvalues_' + r_str, self.d_GN,
                               r'Eigenvalues of $\mathbb{E}_{\nu}[C{\nabla} q^T {\nabla} q]$' + suffix)          # :482-485
            else:
                _save(out, name + self.parameters['output_decoder_name'], mv_to_dense(self.U_NG))
                _save(out, name + '_d_NG', self.d_NG)
                _plot_spectrum(out, name + '_output_eigenvalues_' + r_str, self.d_NG,
                               r'Eigenvalues of $\mathbb{E}_{\nu}[{\nabla} q {\nabla} q^T]$' + suffix)           # :668-671
        return result

    # ---- consumers of the subspaces --------------------------------------------------
    def construct_low_rank_Jacobians(self, check_for_data=True, compress_files=True):
        """Randomized SVDs of this rank's Jacobians for derivative-informed training
        (activeSubspaceProjector.py:690-900: ``hp.accuracyEnhancedSVD(J, Omega_m, parameter_rank, s=1)`` per sample,
        rank ``min(jacobian_rank, q, N)``, dumped as ``J_on_proc{rank}.npz`` with keys U_data / sigma_data / V_data and
        the matching ``mq_on_proc{rank}.npz`` -- ``mzq_on_proc{rank}.npz`` with ``z_data`` for a control problem).  An
        observable speaking the reference's protocol is driven through the reference's loop; otherwise the Jacobians come
        from ``observable.jacobian_data(n)`` with ``n = jacobian_data_per_process`` and the samples from
        ``observable.mq_data(n)`` when it exists.  Returns (U_data, sigma_data, V_data)."""
        if not self.parameters['serialized_sampling']:
            return self._low_rank_jacobians_batched(check_for_data)                                      # :677-678
        return self._low_rank_jacobians(compress_files, parameter_jacobian=True, control_jacobian=False)[0]

    def _low_rank_jacobians_batched(self, check_for_data=True):
        """The batched form (activeSubspaceProjector.py:906-1045): the STORED samples of this rank (``samples_per_process`` of them,
        the ones the subspaces were built from), rank ``min(rank, q, N)``, written as whole arrays into
        ``<output_directory>jacobian_data/``: ``ms_on_proc_{id}.npy``, ``qs_on_proc_{id}.npy`` (``zs_...`` for a control problem),
        ``Us_on_proc_{id}.npy``, ``sigmas_on_proc_{id}.npy``, ``Vs_on_proc_{id}.npy``; existing complete files are returned as they
        are when ``check_for_data`` (the reference resumes sample by sample; here the batch is one device call)."""
        from .datasets import jacobian_svds
        if self.Js is None:
            self._initialize_batched_samples()
        block, ndata, q = self.Js
        rank = min(self.parameters['rank'], q, block.size())                                            # :931
        out = self.parameters['output_directory']
        proc_id = int(self.collective.rank())
        folder = None if out is None else out + 'jacobian_data/'
        names = ('Us', 'sigmas', 'Vs')
        if folder is not None and check_for_data and all(os.path.isfile(folder + n + '_on_proc_' + str(proc_id) + '.npy') for n in names):
            got = tuple(np.load(folder + n + '_on_proc_' + str(proc_id) + '.npy') for n in names)
            if got[0].shape == (ndata, q, rank) and got[1].shape == (ndata, rank):
                return got
        U, sigma, V = jacobian_svds((block, ndata, q), rank)
        if folder is not None:
            os.makedirs(folder, exist_ok=True)
            ms, qs = self.ms, self.qs
            if (ms is None or qs is None) and hasattr(self.observable, 'mq_data'):
                ms, qs = self.observable.mq_data(ndata)
            if ms is not None and qs is not None:
                np.save(folder + 'ms_on_proc_' + str(proc_id) + '.npy', np.asarray(ms))                  # :994-995
                np.save(folder + 'qs_on_proc_' + str(proc_id) + '.npy', np.asarray(qs))
            zs = self._zs_kept if self._zs_kept is not None else self.zs
            if self.control_distribution is not None and zs is not None and zs[0] is not None:
                np.save(folder + 'zs_on_proc_' + str(proc_id) + '.npy',
                        np.asarray([z.get_local() if hasattr(z, 'get_local') else z for z in zs]))       # :998
            np.save(folder + 'Us_on_proc_' + str(proc_id) + '.npy', U)                                   # :1033-1035
            np.save(folder + 'sigmas_on_proc_' + str(proc_id) + '.npy', sigma)
            np.save(folder + 'Vs_on_proc_' + str(proc_id) + '.npy', V)
            _plot_singular_values(out + 'jacobian_singular_values_' + str(rank) + '.pdf', sigma)          # :1041-1042
        return U, sigma, V

    def construct_low_rank_control_Jacobians(self, check_for_data=True, compress_files=True):
        """The same for the Jacobian with respect to the CONTROL variable (activeSubspaceProjector.py:682-688, serialized
        sampling): rank ``min(control_jacobian_rank, q, dim z)``, ``Jz_on_proc{rank}.npz`` with keys Uz_data / sigmaz_data /
        Vz_data next to ``mzq_on_proc{rank}.npz``.  Returns (Uz_data, sigmaz_data, Vz_data)."""
        assert self.control_distribution is not None                                                  # :701
        return self._low_rank_jacobians(compress_files, parameter_jacobian=False, control_jacobian=True)[1]

    def _low_rank_jacobians(self, compress_files, parameter_jacobian, control_jacobian):
        from .datasets import jacobian_svds
        from .operators import ObservableControlJacobian
        n = self.parameters['jacobian_data_per_process']
        obs = self.observable
        mq_pairs = z_data = data = control_data = None
        if not hasattr(obs, 'jacobian_data') and _speaks_reference_protocol(obs):
            # the reference's loop (:726-800): per sample draw, solve, linearise, keep (m, q[, z]), then the Jacobian(s) -- here
            # as dense rows streamed to HBM, instead of hp.accuracyEnhancedSVD applied to the matrix-free Jacobian on the host
            sampler = SeriallySampledJacobianOperator(obs, self.noise, self.prior, operation='JTJ', nsamples=n,
                                                      control_distribution=self.control_distribution, jacobian_factory=ObservableJacobian)
            Jhost = ObservableJacobian(obs) if parameter_jacobian else None
            Jz = ObservableControlJacobian(obs) if control_jacobian else None
            ms, qs, zs, Jz_rows = [], [], [], []

            def points():
                for _ in sampler._points():
                    ms.append(sampler.m.get_local())
                    qs.append(obs.evalu(sampler.u).get_local())
                    if self.control_distribution is not None:
                        zs.append(sampler.z.get_local())
                    if Jz is not None:
                        Jz_rows.append(Jz.dense())
                    yield Jhost.rows() if Jhost is not None else None

            if Jhost is not None:
                q, dM = Jhost.shape
                data = (ingest_stream(points(), n, q, dM, ctx=self.ctx), n, q)
            else:
                for _ in points():
                    pass
            if Jz is not None:
                control_data = np.stack(Jz_rows)
            mq_pairs = (np.stack(ms), np.stack(qs))
            z_data = np.stack(zs) if zs else None
        else:
            if parameter_jacobian:
                data = obs.jacobian_data(n)
            if control_jacobian:
                if not hasattr(obs, 'control_jacobian_data'):
                    raise NotImplementedError("construct_low_rank_control_Jacobians: the observable offers neither the "
                                              "reference's protocol (applyCz / applyCzt ...) nor control_jacobian_data(n)")
                control_data = obs.control_jacobian_data(n)

        def as_block(d):
            if isinstance(d, tuple):
                return d
            d = np.asarray(d, dtype=np.float64)
            nd, rows, cols = d.shape
            return MultiVector.from_vectors(d.reshape(nd * rows, cols), ctx=self.ctx), nd, rows

        out = self.parameters['output_directory']
        save = compress_files and out is not None
        proc_id = int(self.collective.rank())
        if save:
            os.makedirs(out, exist_ok=True)
        results = [None, None]
        ndata = n
        if parameter_jacobian:
            block, ndata, q = as_block(data)
            parameter_rank = min(self.parameters['jacobian_rank'], q, block.size())                 # :724
            results[0] = jacobian_svds((block, ndata, q), parameter_rank)
            if save:
                np.savez_compressed(out + 'J_on_proc' + str(proc_id) + '.npz', U_data=results[0][0], sigma_data=results[0][1],
                                    V_data=results[0][2])                                        # :877-878
                _plot_singular_values(out + 'jacobian_singular_values_' + str(parameter_rank) + '.pdf', results[0][1])   # :880-883
        if control_jacobian:
            block, ndata, q = as_block(control_data)
            wanted = self.parameters['control_jacobian_rank']
            control_rank = min(q, block.size()) if wanted is None else min(wanted, q, block.size())      # :732
            results[1] = jacobian_svds((block, ndata, q), control_rank)
            if save:
                np.savez_compressed(out + 'Jz_on_proc' + str(proc_id) + '.npz', Uz_data=results[1][0], sigmaz_data=results[1][1],
                                    Vz_data=results[1][2])                                       # :896-897
                _plot_singular_values(out + 'control_jacobian_singular_values_' + str(control_rank) + '.pdf', results[1][1])   # :898-901
        if save:
            if mq_pairs is None and hasattr(obs, 'mq_data'):
                mq_pairs = obs.mq_data(ndata)
            if mq_pairs is not None and z_data is not None:
                np.savez_compressed(out + 'mzq_on_proc' + str(proc_id) + '.npz', m_data=mq_pairs[0], z_data=z_data,
                                    q_data=mq_pairs[1])                                          # :864-865
            elif mq_pairs is not None:
                np.savez_compressed(out + 'mq_on_proc' + str(proc_id) + '.npz', m_data=mq_pairs[0], q_data=mq_pairs[1])   # :860
        return results

    def test_errors(self, test_input=True, test_output=False, ranks=[None], cut_off=1e-12, samples=None,
                    output_samples=None):
        """Projection-error tests of the active subspaces (activeSubspaceProjector.py:1037-1230): relative error of
        ``x - V_r V_r^T R x`` (prior-preconditioned input basis; ``V_r V_r^T x`` otherwise, :1093-1096) over parameter
        samples, and of ``q - U_r U_r^T q`` over observable samples.  ``samples`` / ``output_samples`` are blocks or
        (n, dim) arrays; if omitted they come from ``prior.sample_block(n)`` / ``observable.sample_observables(n, ...)``
        with ``n = error_test_samples`` (host PDE-side draws in the reference).  Returns the reference's tuple
        ``(avg_input, std_input)`` / ``(avg_output, std_output)`` / all four, depending on the flags."""
        from .errors import projection_error_test
        want = max((r for r in ranks if r is not None), default=0)
        results = []
        if test_input:
            if self.d_GN is None or len(self.d_GN) < want:
                if want:
                    self.parameters['rank'] = max(self.parameters['rank'], want)
                self.construct_input_subspace()
            if samples is None:
                samples = _prior_sample_block(self.prior, self.parameters['error_test_samples'], self.noise, self.observable)
            B = self.prior.R if self.prior_preconditioned else None
            _, avg, std = projection_error_test(self.V_GN, samples, ranks, B=B, d=self.d_GN, cut_off=cut_off,
                                                collective=self.collective)
            results += [avg, std]
        if test_output:
            if self.d_NG is None or len(self.d_NG) < want:
                if want:
                    self.parameters['rank'] = max(self.parameters['rank'], want)
                self.construct_output_subspace()
            if output_samples is None:
                output_samples = _observable_samples(self.observable, self.prior, self.control_distribution,
                                                     self.parameters['error_test_samples'], self.noise, self.ctx)
            _, avg, std = projection_error_test(self.U_NG, output_samples, ranks, d=self.d_NG, cut_off=cut_off,
                                                collective=self.collective)
            results += [avg, std]
        return tuple(results)


# =====================================================================================
# KLE
# =====================================================================================
class KLEProjector:
    """Input subspace from the prior covariance alone (KLEProjector.py:72-199)."""

    prior_power_iterations = 2      # orthogonality='prior': passes of the randomized generalized solve (see construct_input_subspace)

    def __init__(self, prior, mesh_constructor_comm=None, collective=None, parameters=None, ctx=None):
        self.prior = prior
        self.mesh_constructor_comm = mesh_constructor_comm
        self.collective = collective if collective is not None else NullCollective()
        self.parameters = parameters if parameters is not None else KLEParameterList()
        self.ctx = ctx or L.Context.default()
        self.noise = None
        # the mass matrix: assembled matrices (scipy / PETSc / dolfin) go to HBM as CSR; anything else stays a host
        # operator behind the vector protocol, its size read off the vectors its own init_vector makes
        self.M = as_device_operator(prior.M, ctx=self.ctx)
        self.N = self.M.shape[0]
        if hasattr(prior, "C") and prior.C is not None:
            self.C = as_device_operator(prior.C, self.N, self.ctx)          # explicit covariance (config 2)
        else:
            shaper = H.find_init_vector(prior.Rsolver) or H.find_init_vector(getattr(prior, "R", None)) or H.find_init_vector(prior.M)
            self.C = as_device_operator(Solver2Operator(prior.Rsolver, mpi_comm=mesh_constructor_comm, init_vector=shaper),
                                        self.N, self.ctx)                   # :103
        self.d_KLE = None
        self.V_KLE = None
        self.M_orthogonal = None
        self.R_orthogonal = False

    def _Msolver(self):
        """M^-1 for the M-orthogonal double pass.  With M in HBM as CSR the solve is a device Jacobi-PCG to 1e-13 (tighter
        than the host Krylov solver hippylib builds for ``prior.Msolver``); a mass matrix that could only be wrapped as a
        host operator keeps the prior's own solver."""
        if isinstance(self.M, CsrOperator) and not isinstance(getattr(self.prior, "Msolver", None), DeviceOperator):
            return CsrPCGSolver(self.M.csr, ctx=self.ctx)
        ms = getattr(self.prior, "Msolver", None)
        if ms is None:
            raise ValueError("KLEProjector: prior.Msolver is needed when prior.M is not an assembled matrix")
        return ms

    def random_input_projector(self):
        """A random orthonormal projection basis (:114-128)."""
        Omega = _draw_omega(self.N, self.parameters['rank'] + self.parameters['oversampling'], self.collective, self.ctx)
        Omega.orthogonalize()
        return Omega

    def construct_input_subspace(self, orthogonality='mass'):
        t0 = time.time()
        assert hasattr(self.prior, 'M')
        KLE_Operator = MassPreconditionedCovarianceOperator(self.C, self.M)
        Omega = _draw_omega(self.N, self.parameters['rank'] + self.parameters['oversampling'], self.collective, self.ctx)
        if orthogonality.lower() == 'mass':
            self.d_KLE, self.V_KLE = doublePassG(KLE_Operator, self.M, self._Msolver(), Omega, self.parameters['rank'], s=1)
            self.M_orthogonal = True
            kle_decoder = self.V_KLE
            kle_encoder = MultiVector(kle_decoder)
            MatMvMult(self.M, kle_decoder, kle_encoder)
        elif orthogonality.lower() == 'identity':
            self.d_KLE, self.V_KLE = doublePass(self.C, Omega, self.parameters['rank'], s=1)
            self.M_orthogonal = False
            kle_decoder = self.V_KLE
            kle_encoder = MultiVector(kle_decoder)
        elif orthogonality.lower() == 'prior':
            # The reference hands this mode to SLEPc (KLESubspaceConstructorSLEPc, KLEProjector.py:285-334): Krylov-Schur with
            # shift-and-invert on A v = mu M v (R = A M^-1 A), decoder_i = v_i / mu_i, eigenvalues 1 / mu_i^2, encoder = R decoder.
            # Those are the dominant eigenpairs of  M u = lambda R u  with u^T R u = 1  (lambda = 1 / mu^2, u = v / mu), i.e. a
            # generalized problem of exactly the kind this library solves: doublePassG(A = M, B = R, B^-1 = Rsolver).  It is the
            # RANDOMIZED method, not Krylov-Schur: `self.prior_power_iterations` passes (attribute, default 2: the covariance spectrum
            # of a bi-Laplacian prior decays algebraically) stand in for SLEPc's convergence test -- more passes, more accuracy.  (An
            # attribute, not a parameter key: the parameter lists keep exactly the reference's keys.)
            assert hasattr(self.prior, 'R') and hasattr(self.prior, 'Rsolver')
            R_op = as_device_operator(self.prior.R, self.N, self.ctx)
            self.d_KLE, self.V_KLE = doublePassG(self.M, R_op, self.C, Omega, self.parameters['rank'],
                                                 s=max(1, int(self.prior_power_iterations)))
            self.M_orthogonal = False
            self.R_orthogonal = True
            kle_decoder = self.V_KLE
            kle_encoder = MultiVector(kle_decoder)
            MatMvMult(R_op, kle_decoder, kle_encoder)                                             # :333
        else:
            raise ValueError(orthogonality)
        self._subspace_construction_time = time.time() - t0
        if self.parameters['verbose'] and _is_root(self.collective):
            print('Construction of input subspace took ', self._subspace_construction_time, 's')
        if _is_root(self.collective) and self.parameters['save_and_plot'] and self.parameters['output_directory'] is not None:
            _save(self.parameters['output_directory'], self.parameters['input_decoder_name'], mv_to_dense(self.V_KLE))
            _save(self.parameters['output_directory'], 'KLE_d', self.d_KLE)
            _plot_spectrum(self.parameters['output_directory'], 'KLE_eigenvalues_' + str(self.parameters['rank']), self.d_KLE,
                           r'Eigenvalues of $C$' + self.parameters['plot_label_suffix'])                              # :194-197
        return self.d_KLE, kle_decoder, kle_encoder


    def test_errors(self, ranks=[None], cut_off=1e-12, samples=None):
        """Projection-error test of the KLE basis (KLEProjector.py:202-282).  ``samples`` is a block / (n, N) array
        of prior draws; if omitted ``prior.sample_block(n)`` supplies ``error_test_samples`` of them (the
        reference loops prior.sample over hp.parRandom noise, a host PDE-side operation)."""
        from .errors import projection_error_test
        want = max((r for r in ranks if r is not None), default=0)
        if self.d_KLE is None or len(self.d_KLE) < want:
            if want:
                self.parameters['rank'] = max(self.parameters['rank'], want)
            self.construct_input_subspace()
        if samples is None:
            samples = _prior_sample_block(self.prior, self.parameters['error_test_samples'])
        B = self.M if self.M_orthogonal else (as_device_operator(self.prior.R, self.N, self.ctx) if self.R_orthogonal else None)
        _, avg, std = projection_error_test(self.V_KLE, samples, ranks, B=B, d=self.d_KLE, cut_off=cut_off, collective=self.collective)
        return avg, std


class BoundaryRestrictedKLEProjector:
    """Prior-based KLE subspace for boundary data (KLEProjector.py:336-435): the generalized eigenproblem
    ``M_b C M_b u = lambda B u`` with the boundary-restricted mass matrix ``M_b`` (singular: zero rows for interior
    nodes) and its invertible completion ``B = M_b + I_interior``; decoder B-orthonormal, encoder ``M_b * decoder``.

    The reference assembles ``M_b`` with FEniCS from the boundary measure ``ds`` (:364-396); here the assembled matrix
    is handed over (``boundary_mass``, a scipy sparse matrix, or ``prior.M_boundary``) and ``ds`` is accepted only for
    signature parity.  ``B`` follows the reference's rule (:382-394): a one on every diagonal entry of ``M_b`` that is
    numerically zero.  ``B^-1`` is the reference's MUMPS LU (:360-361); on the device it is a Jacobi-PCG solve (B is
    a boundary mass matrix plus an identity block: well conditioned)."""

    def __init__(self, prior, ds=None, parameters=None, boundary_mass=None, ctx=None):
        import scipy.sparse as sp
        self.prior = prior
        self.ds = ds
        self.parameters = parameters if parameters is not None else KLEParameterList()
        self.ctx = ctx or L.Context.default()
        if boundary_mass is None:
            boundary_mass = getattr(prior, "M_boundary", None)
        if boundary_mass is None:
            raise ValueError("BoundaryRestrictedKLEProjector: pass the assembled boundary mass matrix "
                             "(boundary_mass=... or prior.M_boundary); FEniCS assembly is a host-side black box")
        self._Mb_csr = sp.csr_matrix(boundary_mass)
        self.N = self._Mb_csr.shape[0]
        self.M = CsrOperator(self.make_boundary_restricted_mass_matrix(fill_nullspace=False), ctx=self.ctx)
        self._B_csr = self.make_boundary_restricted_mass_matrix(fill_nullspace=True)
        self.B = CsrOperator(self._B_csr, ctx=self.ctx)
        if hasattr(prior, "C") and prior.C is not None:
            self.C = as_device_operator(prior.C, self.N, self.ctx)
        else:
            self.C = as_device_operator(Solver2Operator(prior.Rsolver), self.N, self.ctx)    # :356
        self.KLE_Operator = MassPreconditionedCovarianceOperator(self.C, self.M)              # :357
        self.Bsolver = CsrPCGSolver(self._B_csr, rel_tol=1e-13, ctx=self.ctx)                 # :360-361
        self.d_KLE = None
        self.V_KLE = None

    def make_boundary_restricted_mass_matrix(self, fill_nullspace=False):
        """The boundary mass matrix, optionally with the reference's nullspace fill (:364-396)."""
        import scipy.sparse as sp
        Mb = self._Mb_csr
        if not fill_nullspace:
            return Mb
        new_diag = np.isclose(Mb.diagonal(), 0.0).astype(np.float64)
        return (Mb + sp.diags(new_diag)).tocsr()

    def construct_input_subspace(self):
        """(d, decoder, encoder) of the boundary-restricted KLE (:399-435)."""
        rank = self.parameters['rank']
        oversampling = self.parameters['oversampling']
        Omega = MultiVector(self.N, rank + oversampling, ctx=self.ctx)
        parRandom.normal(1.0, Omega)                                                          # :423-425
        self.d_KLE, self.V_KLE = doublePassG(self.KLE_Operator, self.B, self.Bsolver, Omega, rank, s=1)   # :428
        KLE_encoder = MultiVector(self.N, rank, ctx=self.ctx)
        MatMvMult(self.M, self.V_KLE, KLE_encoder)                                            # :431-432
        return self.d_KLE, self.V_KLE, KLE_encoder


# =====================================================================================
# POD
# =====================================================================================
class PODProjector:
    """Output projector from sampled observables: dominant eigenpairs of E[q q^T] by a randomized
    double pass over the snapshot-Gram operator (PODProjector.py:331-389)."""

    def __init__(self, observable, prior, control_distribution=None, mesh_constructor_comm=None, collective=None,
                 parameters=None, ctx=None):
        self.parameters = parameters if parameters is not None else PODParameterList()
        self.observable = observable
        self.prior = prior
        self.control_distribution = control_distribution
        if mesh_constructor_comm is None and hasattr(observable, "mpi_comm"):
            mesh_constructor_comm = observable.mpi_comm()                               # :66-69
        self.mesh_constructor_comm = mesh_constructor_comm
        self.collective = collective if collective is not None else NullCollective()   # the reference forgets the import (:77)
        self.ctx = ctx or L.Context.default()
        self.noise = None
        if _speaks_reference_protocol(observable) and hasattr(prior, "init_vector"):
            self.noise = H.new_host_vector(mesh_constructor_comm)                       # :84-85
            prior.init_vector(self.noise, "noise")
        self.d = None
        self.U_MV = None
        self.u_at_mean = None
        self.LocalObservables = None

    def set_snapshots(self, snapshots):
        """Precomputed local snapshots: (n, N) array (q_data layout, :224-225) or a MultiVector."""
        self.LocalObservables = snapshots if isinstance(snapshots, MultiVector) else MultiVector.from_vectors(snapshots, ctx=self.ctx)

    def solve_at_mean(self):
        """The forward solve at the prior mean (:99-113): sets up a nonlinear problem before the sampling loop."""
        problem = getattr(self.observable, "problem", None)
        if problem is None or not hasattr(self.prior, "mean"):
            return
        self.u_at_mean = problem.generate_state()
        point = [self.u_at_mean, self.prior.mean, None]
        if self.control_distribution is not None:
            if hasattr(self.control_distribution, 'mean'):
                point.append(self.control_distribution.mean)
            else:
                z_mean = self.observable.generate_vector(H.CONTROL)
                self.control_distribution.sample(z_mean)
                point.append(z_mean)
        problem.solveFwd(self.u_at_mean, point)

    def construct_subspace(self):
        t0 = time.time()
        if self.LocalObservables is None:
            if _speaks_reference_protocol(self.observable):
                self.solve_at_mean()                                                          # :336
            self.LocalObservables = _observable_samples(self.observable, self.prior, self.control_distribution,
                                                        self.parameters['sample_per_process'], self.noise, self.ctx)   # :343-357
        X = self.LocalObservables
        LocalPODOperator = SnapshotGramOperator(X, scale=1.0 / X.nvec())                      # :359-361
        GlobalPODOperator = CollectiveOperator(LocalPODOperator, self.collective, mpi_op='avg')  # :363
        Omega_POD = _draw_omega(X.size(), self.parameters['rank'] + self.parameters['oversampling'], self.collective, self.ctx)
        self.d, self.U_MV = doublePass(GlobalPODOperator, Omega_POD, self.parameters['rank'], s=1)   # :376
        self._subspace_construction_time = time.time() - t0
        if self.parameters['verbose'] and _is_root(self.collective):
            print('Construction of POD subspace took ', self._subspace_construction_time, 's')
        if _is_root(self.collective) and self.parameters['output_directory'] is not None:
            _save(self.parameters['output_directory'], 'POD_projector', mv_to_dense(self.U_MV))
            _save(self.parameters['output_directory'], 'POD_d', self.d)
            _plot_spectrum(self.parameters['output_directory'], 'POD_eigenvalues_' + str(self.parameters['rank']), self.d,
                           r'Eigenvalues of $\mathbb{E}_{\nu}[qq^T]$' + self.parameters['plot_label_suffix'])          # :386-389

    def test_output_errors(self, ranks=[None], cut_off=1e-10, samples=None):
        """Projection-error test on the output (PODProjector.py:392-478): relative error of projecting observable
        samples onto the leading POD vectors.  ``samples`` defaults to fresh draws from the observable."""
        from .errors import projection_error_test
        if self.d is None or self.U_MV is None:
            self.construct_subspace()
        if samples is None:
            samples = _observable_samples(self.observable, self.prior, self.control_distribution,
                                          self.parameters['sample_per_process'], self.noise, self.ctx)
        _, avg, std = projection_error_test(self.U_MV, samples, ranks, d=self.d, cut_off=cut_off, collective=self.collective)
        return avg, std


    def generate_training_data(self, check_for_data=True, sequential=True, compress_files=True):
        """``data_per_process`` (m, q) pairs of this rank in the reference's on-disk forms, resumable
        (PODProjector.py:118-297; ``datasets.generate_training_data``)."""
        from .datasets import generate_training_data
        self.solve_at_mean()
        t0 = time.time()
        made = generate_training_data(self.observable, self.prior, self.parameters['data_per_process'], self.parameters['output_directory'],
                                      noise=self.noise, control_distribution=self.control_distribution, rank=int(self.collective.rank()),
                                      check_for_data=check_for_data, sequential=sequential, compress_files=compress_files,
                                      u_init=None if sequential else self.u_at_mean)
        self._data_generation_time = time.time() - t0
        return made

    def input_output_error_test(self, V_MV, Cinv=None, rank_pairs=[None]):
        """Input-output projection error test (PODProjector.py:541-655): the output basis is this projector's POD basis,
        ``V_MV`` the input basis (an AS decoder with ``Cinv = prior.R``, a KLE decoder with ``Cinv = prior.M``, or a random
        one), ``rank_pairs`` the (input rank, output rank) pairs.  Returns (global_avg_rel_errors, global_std_rel_errors)."""
        from .errors import input_output_error_test
        assert self.d is not None and self.U_MV is not None
        return input_output_error_test(self.observable, self.prior, self.U_MV, V_MV, rank_pairs, self.parameters['sample_per_process'],
                                       Cinv=Cinv, noise=self.noise, collective=self.collective,
                                       control_distribution=self.control_distribution)


def weighted_l2_norm_vector(x, W):
    """PODProjector.py:658-661."""
    Wx = W @ x
    norm2 = np.einsum('ij,ij->j', Wx, x)
    return np.sqrt(norm2)


class PODProjectorFromData:
    """Deterministic mass-weighted POD from a snapshot matrix (PODProjector.py:666-852).
    ``method='hep'`` (n << N) runs on the device: the n x n Gram matrix X^T M X and the back-transform
    phi = X U are tall-skinny contractions, the n x n symmetric eigensolve is tridiagonalisation + divide and conquer
    (one workgroup up to 256 snapshots; panels, merges and block reflectors over the whole GPU up to 16384:
    ``hfmi_block_gram_eig``, only the u_rank wanted eigenvectors come back) -- same steps as :812-833.

    ``method='ghep'`` (H = M X (M X)^T / n against M, :743-773) and ``'inverse_ghep'`` (H = X X^T / n against
    M^-1, :775-810) are ARPACK Lanczos iterations on the host in the reference.  Both pencils have their
    eigenvectors in range(X): with phi = X c they reduce to the SAME n x n problem (X^T M X) c = n lambda c that
    'hep' solves, with the same normalisation phi^T M phi = 1 (ghep: eigsh's M-orthonormality; inverse_ghep:
    (M phi)^T M^-1 (M phi) = 1).  They are therefore served by the device Gram route too; results agree with the
    reference's Lanczos output up to the sign of each mode and its iteration tolerance (tests/golden)."""

    def __init__(self, Vh=None, M_output=None, ctx=None):
        import scipy.sparse as sp
        self.Vh = Vh
        if M_output is None:
            raise ValueError("PODProjectorFromData: pass the output mass matrix as a scipy sparse matrix "
                             "(the reference assembles it with FEniCS, PODProjector.py:681-690)")
        self.M_csr = sp.csr_matrix(M_output)
        self.ctx = ctx or L.Context.default()

    EXACT_MAX_SNAPSHOTS = 16384          # HFMI_EIG_MAXN: the largest n x n eigensolve on the device
    prefer_state_dimension = True        # N x N form of the pencil when the state dimension is at most half the number of snapshots
    RANDOMIZED_PASSES = 3                # power passes of the fallback beyond it
    RANDOMIZED_OVERSAMPLING = 40

    def _randomized(self, u_data, u_rank, oversampling=None, passes=None):
        """More than 16384 snapshots AND a state dimension beyond 16384 (neither the n x n nor the N x N problem fits the exact
        eigensolver; the reference's :812-833 takes any n): the same modes from the N-dimensional form of the problem, (1/n) M X X^T M phi = lambda M phi with phi^T M phi = 1, by the
        randomized double pass this library is built around (``doublePassG`` with B = M, B^-1 = the device mass solve) with
        ``oversampling`` = 40 extra probe columns and ``passes`` = 3 applications of the operator per side.

        STATED TOLERANCE (tests/test_gpu_solvers.py::test_pod_from_data_randomized_fallback_tolerance): with k = u_rank + 40 probe
        columns and s = 3 passes the relative error of eigenvalue i is bounded by about (lambda_{k+1} / lambda_i)^(2 s - 1)
        (Halko-Martinsson-Tropp 2011, section 10.4) -- exact to rounding when the snapshots' numerical rank is <= k, <= 1e-6 for
        the leading u_rank modes as soon as lambda_{k+1} <= 0.06 lambda_{u_rank}; a spectrum flatter than that over 40 modes is
        reported by the warning below with the measured ratio, so that the caller can raise the oversampling."""
        import warnings
        from .operators import ComposedOperator
        oversampling = self.RANDOMIZED_OVERSAMPLING if oversampling is None else oversampling
        passes = self.RANDOMIZED_PASSES if passes is None else passes
        X = MultiVector.from_vectors(u_data, ctx=self.ctx)
        Mop = CsrOperator(self.M_csr, ctx=self.ctx)
        A = ComposedOperator(Mop, SnapshotGramOperator(X, scale=1.0 / X.nvec()), Mop)
        k = min(u_rank + oversampling, X.size())
        Omega = _draw_omega(X.size(), k, NullCollective(), self.ctx)
        d_all, phi_all = doublePassG(A, Mop, CsrPCGSolver(Mop.csr, ctx=self.ctx), Omega, k, s=passes)
        tail = float(d_all[-1] / d_all[u_rank - 1]) if d_all[u_rank - 1] > 0 else 0.0
        warnings.warn("PODProjectorFromData: %d snapshots > %d -- randomized double pass (k = %d probe columns, %d passes) on the "
                      "N-dimensional generalized problem instead of the n x n Gram eigensolve; lambda_k / lambda_r = %.2e, estimated "
                      "relative eigenvalue error of mode r <= %.1e" % (u_data.shape[0], self.EXACT_MAX_SNAPSHOTS, k, passes, tail,
                                                                      tail ** (2 * passes - 1)))
        d = d_all[:u_rank]
        phi_mv = phi_all if k == u_rank else MultiVector.from_vectors(np.ascontiguousarray(phi_all.to_dense()[:, :u_rank].T), ctx=self.ctx)
        Mphi_mv = MultiVector(phi_mv)
        Mop.matMvMult(phi_mv, Mphi_mv)
        return d, phi_mv.to_dense(), Mphi_mv.to_dense()

    def _state_dimension_route(self, u_data, u_rank):
        """More snapshots than the n x n eigensolver takes, but a state dimension N it does take (the usual large case: the POD of an
        OUTPUT of a few hundred or thousand observables over a training set of tens of thousands, dataGenerator.py:278-279): the same
        eigenpairs, exactly, from the N-dimensional form.  With M = B B^T (B = Q_M diag(sqrt(lambda_M)) from the eigendecomposition of
        the mass matrix) and phi = B^-T y the pencil (1/n) M X^T X M phi = lambda M phi becomes the symmetric N x N problem
        (B^T (X^T X) B) y = n lambda y, phi^T M phi = y^T y = 1.  Every product and both eigensolves run on the device:
        X^T X = ``dot_mv`` of the N columns of the snapshot matrix, the two N x N x N congruence products = ``hfmi_dense_matmul``."""
        n, N = u_data.shape
        if u_rank > N:
            raise ValueError("PODProjectorFromData: rank %d exceeds the state dimension %d (the snapshots span at most that many modes)" % (u_rank, N))
        Xt = MultiVector.from_vectors(np.ascontiguousarray(u_data.T), ctx=self.ctx)     # N vectors of length n: the columns of u_data
        XtX = Xt.dot_mv(Xt)
        if getattr(self, "_mass_factor", None) is None:          # M = Q diag(lam_M) Q^T, once per projector (the mass matrix is fixed)
            lam_M, Q = sym_eig_small(self.M_csr.toarray(), ctx=self.ctx)
            if not lam_M[-1] > 0.0:
                raise ValueError("PODProjectorFromData: the output mass matrix is not positive definite (smallest eigenvalue %.3e)" % lam_M[-1])
            self._mass_factor = (lam_M, Q)
        lam_M, Q = self._mass_factor
        root = np.sqrt(lam_M)
        B = Q * root[None, :]
        S = self.ctx.dense_matmul(B, self.ctx.dense_matmul(XtX, B), ta=True)
        mu, Y = sym_eig_small(0.5 * (S + S.T), ctx=self.ctx, nvec=u_rank)
        phi = self.ctx.dense_matmul(Q / root[None, :], np.ascontiguousarray(Y[:, :u_rank]))
        phi_mv = MultiVector.from_dense(np.ascontiguousarray(phi), ctx=self.ctx)
        Mphi_mv = MultiVector(phi_mv)
        Mop = CsrOperator(self.M_csr, ctx=self.ctx)
        Mop.matMvMult(phi_mv, Mphi_mv)
        norms = np.sqrt(np.diag(phi_mv.dot_mv(Mphi_mv)))                 # weighted_l2_norm_vector, :829 (1 to rounding here)
        phi = phi / norms[None, :]
        return mu[:u_rank] / n, phi, Mphi_mv.to_dense() / norms[None, :]

    def construct_subspace(self, u_data, u_rank, shifted=True, method='hep', verify=False):
        n_data, dim_u = u_data.shape
        assert u_rank <= n_data, "number of samples needs to be greater than rank of projector"
        if shifted:
            u_shift = np.mean(u_data, axis=0)
            u_data = u_data - u_shift
        else:
            u_shift = np.zeros(u_data.shape[1])
        # the same pencil in whichever of its two exact forms is the smaller problem: n x n (the reference's la.eigh(G)) or N x N
        # (the state dimension: the POD of an output of a few hundred observables over thousands of samples is 8300 x 600 -> a 600 x 600
        # problem, 15 ms instead of 270).  ``prefer_state_dimension = False`` keeps the n x n form whenever it fits.
        small_state = dim_u <= self.EXACT_MAX_SNAPSHOTS and u_rank <= dim_u and (
            n_data > self.EXACT_MAX_SNAPSHOTS or (self.prefer_state_dimension and 2 * dim_u <= n_data))
        if method in ('hep', 'ghep', 'inverse_ghep') and small_state:
            d, phi, Mphi = self._state_dimension_route(u_data, u_rank)                  # exact: N x N instead of n x n
        elif method in ('hep', 'ghep', 'inverse_ghep') and n_data > self.EXACT_MAX_SNAPSHOTS:
            d, phi, Mphi = self._randomized(u_data, u_rank)                               # both n and N beyond the exact solver
        elif method in ('hep', 'ghep', 'inverse_ghep'):
            X = MultiVector.from_vectors(u_data, ctx=self.ctx)        # one snapshot per vector
            Mop = CsrOperator(self.M_csr, ctx=self.ctx)
            MX = MultiVector(X.size(), X.nvec(), ctx=self.ctx)
            Mop.matMvMult(X, MX)
            s, U = X.gram_eig(MX, u_rank)                              # :818-823: UtMU, eigh, descending; only U[:, :u_rank] is used
            d = s[:u_rank] / n_data
            U = np.ascontiguousarray(U[:, :u_rank])
            from .multivector import MvDSmatMult
            phi_mv = MultiVector(X.size(), u_rank, ctx=self.ctx)
            Mphi_mv = MultiVector(X.size(), u_rank, ctx=self.ctx)
            MvDSmatMult(X, U, phi_mv)                                  # :826
            Mop.matMvMult(phi_mv, Mphi_mv)
            norms = np.sqrt(np.diag(phi_mv.dot_mv(Mphi_mv)))           # weighted_l2_norm_vector, :829
            MvDSmatMult(X, np.ascontiguousarray(U / norms), phi_mv)    # phi / ||phi||_M
            Mop.matMvMult(phi_mv, Mphi_mv)                             # :830
            phi, Mphi = phi_mv.to_dense(), Mphi_mv.to_dense()
        else:
            raise ValueError("Unavailable method")
        if verify:
            r = u_rank - 1 if shifted else u_rank
            phi_orth_error = np.linalg.norm(phi[:, :r].T @ self.M_csr @ phi[:, :r] - np.eye(r))
            print(f"Basis Orthogonality error: {phi_orth_error}")
        return d, phi, Mphi, u_shift
