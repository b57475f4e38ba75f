"""Training-data generation for derivative-informed surrogates: the caller on the far side of the hot path
(SURVEY.md section 8f ranks 1 and 3).

Reference: ``DataGenerator`` and ``compress_dataset`` (modeling/dataGenerator.py:25-700).  The PDE work stays the host's, behind
the reference's own observable / prior protocol: per sample a prior draw (and a control draw), ``solveFwd``,
``setLinearizationPoint``, ``evalu``.  What the device takes over is everything that is dense linear algebra on the sample's
Jacobian -- ``J^T (M Phi)``, ``J Psi``, or the randomized SVD ``hp.accuracyEnhancedSVD(J, Omega, r, s=1)`` -- whenever
materialising the Jacobian (min(q, dM) incremental solves through ``ObservableJacobian.dense``) costs no more solves than the
matrix-free route would: the rows of a whole chunk of samples are streamed to HBM through pinned buffers while the host solves the
next sample (``multivector.ingest_stream``), and the chunk is contracted / factorised there in one go.  Otherwise (a full-state
observable against a small reduced basis) the matrix-free columns ARE the result and nothing is uploaded.

On disk the contract is the reference's, name for name and key for key: per-sample ``mq_data/{m,q}_sample_i.npy``
(``mzq_data/`` with ``z_sample_i.npy`` for control problems), ``J_data/{JstarPhi,JPsi}i.npy`` or ``{U,sigma,V}_sample_i.npy``,
``Jz_data/JzstarPhi i.npy`` or ``{Uz,sigmaz,Vz}_sample_i.npy``, ``skipped/`` for draws whose forward solve failed; after
``compress_dataset``: ``mq_data.npz`` / ``mzq_data.npz``, ``JstarPhi_data.npz``, ``JPsi_data.npz``, ``Jsvd_data.npz`` and their
``Jz`` twins (dataGenerator.py:634-655).
"""
import os
import shutil

import numpy as np

from . import hostvec as H
from .multivector import MultiVector, ingest_stream
from .operators import ObservableControlJacobian, ObservableJacobian, StateSpaceIdentityOperator


def data_generator_settings(settings=None):
    """dataGenerator.py:25-35: ranks of the randomized SVDs of J (``rM``) and Jz (``rZ``), their oversampling, whether the
    state is zeroed before every solve, whether draws with a failed solve are kept, verbosity."""
    settings = {} if settings is None else settings
    settings['rM'] = None
    settings['rZ'] = None
    settings['oversample'] = 10
    settings['reset_initial_guess'] = False
    settings['save_failed_solves'] = True
    settings['verbose'] = True
    return settings


def _dir(*parts):
    path = os.path.join(*parts)
    os.makedirs(path, exist_ok=True)
    return path + os.sep


class _DerivativeSink:
    """What is kept of one Jacobian (parameter or control) per sample, and by which route.

    ``kind``: 'adjoint_basis' (J^T MPhi), 'forward_basis' (J Psi) or 'svd'.  ``on_device``: the Jacobian is materialised and the
    work done in HBM per chunk of samples; otherwise the result comes column by column from the matrix-free Jacobian on the host."""

    def __init__(self, J, kind, basis, rank, oversample, folder, stems, ctx, stream_rows):
        self.J, self.kind, self.basis, self.ctx = J, kind, basis, ctx
        self.folder, self.stems = folder, stems
        self.stream_rows = stream_rows          # the big (parameter) Jacobian goes through ingest_stream, the small one is stacked
        q, n = J.shape
        dense_solves = min(q, n)
        if kind == 'svd':
            self.nvec = min(rank + oversample, q, n)                  # dataGenerator.py:411,448
            self.rank = min(rank, self.nvec)
            free_solves = 4 * self.nvec                               # Y = A Omega, A^T Y, A Z, A^T Q
        else:
            self.rank = basis.shape[1]
            free_solves = self.rank
        self.on_device = dense_solves <= free_solves
        self._held = []

    # ---- per sample, right after the linearisation
    def take(self, index):
        """Returns the dense Jacobian when it is to be streamed by the caller, else None."""
        if self.on_device:
            dense = self.J.dense()
            if self.stream_rows:
                return dense
            self._held.append(dense)
            return None
        self._write(index, self._matrix_free())
        return None

    def _matrix_free(self):
        J, comm = self.J, self.J.mpi_comm()
        q, n = J.shape
        if self.kind == 'svd':
            from .randomized import accuracyEnhancedSVD, parRandom
            op = _HostRectangular(J, self.ctx)
            Omega = MultiVector(n, self.nvec, ctx=self.ctx)
            parRandom.normal(1.0, Omega)
            U, sigma, V = accuracyEnhancedSVD(op, Omega, self.rank, s=1)
            return U.to_dense(), sigma, V.to_dense()
        adjoint = self.kind == 'adjoint_basis'
        src = H.shape_with(J.init_vector, 0 if adjoint else 1, comm)
        dst = H.shape_with(J.init_vector, 1 if adjoint else 0, comm)
        out = np.empty((n if adjoint else q, self.rank))
        for j in range(self.rank):
            src.set_local(np.ascontiguousarray(self.basis[:, j]))
            src.apply("")
            (J.transpmult if adjoint else J.mult)(src, dst)
            out[:, j] = dst.get_local()
        return (out,)

    # ---- per chunk
    def flush(self, first, count, block=None):
        if not self.on_device or count == 0:
            return
        from .datasets import jacobian_svds, jacobian_times_input_basis, jacobian_transpose_times_output_basis
        q, n = self.J.shape
        if block is None:
            block = MultiVector.from_vectors(np.stack(self._held).reshape(count * q, n), ctx=self.ctx)
            self._held = []
        data = (block, count, q)
        if self.kind == 'adjoint_basis':
            results = (jacobian_transpose_times_output_basis(data, self.basis),)
        elif self.kind == 'forward_basis':
            results = (jacobian_times_input_basis(data, self.basis),)
        else:
            results = jacobian_svds(data, self.rank, oversampling=self.nvec - self.rank)
        for i in range(count):
            self._write(first + i, tuple(r[i] for r in results))

    def _write(self, index, arrays):
        for stem, a in zip(self.stems[self.kind], arrays):
            np.save(self.folder + stem % index + '.npy', a)


class _HostRectangular:
    """A host Jacobian (``mult`` / ``transpmult`` on host vectors) behind device vectors, column by column: what lets the device's
    randomized SVD drive a matrix-free operator."""

    def __init__(self, J, ctx):
        self.J, self.ctx = J, ctx
        comm = J.mpi_comm()
        self._range, self._domain = H.shape_with(J.init_vector, 0, comm), H.shape_with(J.init_vector, 1, comm)

    def init_vector(self, x, dim):
        x.init(self.J.shape[0] if dim == 0 else self.J.shape[1])

    def _apply(self, fn, src, dst, x, y):
        src.set_local(x.get_local())
        src.apply("")
        fn(src, dst)
        y.set_local(dst.get_local())

    def mult(self, x, y):
        self._apply(self.J.mult, self._domain, self._range, x, y)

    def transpmult(self, x, y):
        self._apply(self.J.transpmult, self._range, self._domain, x, y)


_J_STEMS = {'adjoint_basis': ('JstarPhi%d',), 'forward_basis': ('JPsi%d',), 'svd': ('U_sample_%d', 'sigma_sample_%d', 'V_sample_%d')}
_JZ_STEMS = {'adjoint_basis': ('JzstarPhi%d',), 'forward_basis': ('JzPsi%d',),
             'svd': ('Uz_sample_%d', 'sigmaz_sample_%d', 'Vz_sample_%d')}


class DataGenerator:
    """dataGenerator.py:37-493.  ``generate`` samples the parameter-to-observable map (and its derivatives),
    ``compute_jacobians_in_subspace`` re-linearises at stored samples, ``two_step_generate`` chains both around a POD of the
    sampled states."""

    chunk_bytes = 8 << 30          # Jacobian rows held in HBM per chunk of samples

    def __init__(self, observable, prior, control_distribution=None, settings=None, parRandom=None, mesh_constructor_comm=None,
                 ctx=None):
        from . import _lib as L
        from .randomized import parRandom as process_wide
        self.observable, self.prior, self.control_distribution = observable, prior, control_distribution
        self.mesh_constructor_comm = mesh_constructor_comm if mesh_constructor_comm is not None else observable.mpi_comm()
        self.parRandom = process_wide if parRandom is None else parRandom
        self.settings = data_generator_settings() if settings is None else settings
        self.ctx = ctx or L.Context.default()
        self.noise = H.new_host_vector(self.mesh_constructor_comm)
        prior.init_vector(self.noise, "noise")
        self.u = self.m = self.z = None
        self.J = self.Jz = None
        self.dQ = self.dM = self.dZ = None
        self.exceptions_count = 0

    # ---- set-up shared by the entry points (dataGenerator.py:360-493)
    def initialize_sampling(self, derivatives, input_decoder=None, input_encoder=None, output_decoder=None, output_encoder=None):
        obs, gen = self.observable, self.observable.generate_vector
        self.u = gen(H.STATE) if self.u is None else self.u
        self.m = gen(H.PARAMETER) if self.m is None else self.m
        if self.control_distribution is not None and self.z is None:
            self.z = gen(H.CONTROL)
        if getattr(obs.problem, 'C', True) is None:
            # the KKT blocks do not exist before a first linearisation: solve at the mean (dataGenerator.py:467-486)
            point = [obs.problem.generate_state(), self.prior.mean, None]
            if self.control_distribution is not None:
                if hasattr(self.control_distribution, 'mean'):
                    point.append(self.control_distribution.mean)
                else:
                    z_mean = gen(H.CONTROL)
                    self.control_distribution.sample(z_mean)
                    point.append(z_mean)
            obs.problem.solveFwd(point[0], point)
            obs.setLinearizationPoint(point)
        if output_decoder is not None and output_encoder is None:
            if self.settings['verbose']:
                print('DataGenerator: no output encoder given -- using the decoder (right only for the Euclidean inner product)')
            output_encoder = output_decoder
        sinks = {}
        oversample = self.settings['oversample']
        if derivatives[0]:
            self.J = ObservableJacobian(obs)
            self.dQ, self.dM = self.J.shape
            if output_decoder is not None:
                kind, basis, rank = 'adjoint_basis', np.asarray(output_encoder, dtype=np.float64), None
            elif input_decoder is not None:
                kind, basis, rank = 'forward_basis', np.asarray(input_decoder, dtype=np.float64), None
            else:
                kind, basis, rank = 'svd', None, self.settings['rM']
                assert rank is not None, "settings['rM']: the rank of the Jacobian's randomized SVD"
            sinks['J'] = (kind, basis, rank, oversample, _J_STEMS, True)
        if derivatives[1]:
            assert self.control_distribution is not None
            self.Jz = ObservableControlJacobian(obs)
            self.dQ, self.dZ = self.Jz.shape
            if output_decoder is not None:
                kind, basis, rank = 'adjoint_basis', np.asarray(output_encoder, dtype=np.float64), None
            else:                                  # an input basis lives in the parameter space: the control Jacobian is factorised
                kind, basis, rank = 'svd', None, self.settings['rZ']
                assert rank is not None, "settings['rZ']: the rank of the control Jacobian's randomized SVD"
            sinks['Jz'] = (kind, basis, rank, oversample, _JZ_STEMS, False)
        if self.dM is None:
            self.dM = len(self.m.get_local())
        return sinks

    def _make_sinks(self, plan, data_dir):
        made = {}
        for name, (kind, basis, rank, oversample, stems, stream_rows) in plan.items():
            J = self.J if name == 'J' else self.Jz
            made[name] = _DerivativeSink(J, kind, basis, rank, oversample, _dir(data_dir, name + '_data'), stems, self.ctx,
                                         stream_rows)
        return made

    def _chunk(self, sinks, n_samples):
        big = sinks.get('J')
        if big is None or not big.on_device:
            return n_samples
        return max(1, min(n_samples, int(self.chunk_bytes // (8 * self.dQ * self.dM))))

    def _run_chunks(self, n_samples, sinks, points):
        """``points(first, count, materialise)`` linearises one sample after the other, calls ``materialise(index)`` -- the
        derivative work of that sample: dense Jacobian rows or the matrix-free products -- INSIDE whatever retry scope it has
        (dataGenerator.py:125-239 wraps the derivative computation in the same try as the forward solve and redraws the
        sample), and yields what it returned."""
        done = 0
        chunk = self._chunk(sinks, n_samples)
        big, small = sinks.get('J'), sinks.get('Jz')

        def materialise(index):
            held = len(small._held) if small is not None else 0
            try:
                if small is not None:
                    small.take(index)
                return big.take(index) if big is not None else None
            except Exception:
                if small is not None:
                    del small._held[held:]          # the sample is drawn again: nothing of this attempt may stay queued
                raise

        while done < n_samples:
            count = min(chunk, n_samples - done)

            def rows():
                yield from points(done, count, materialise)

            if big is not None and big.on_device:
                block = ingest_stream(rows(), count, self.dQ, self.dM, ctx=self.ctx)
                big.flush(done, count, block)
            else:
                for _ in rows():
                    pass
            if small is not None:
                small.flush(done, count)
            done += count

    # ---- m -> q(m) [and derivatives] over fresh draws (dataGenerator.py:88-248)
    def generate(self, n_samples, derivatives=(0, 0), output_decoder=None, input_decoder=None, n_data_per_sample=1,
                 data_dir='data/test/', compress=True, clean_up=True, output_encoder=None, input_encoder=None):
        control = self.control_distribution is not None
        data_dir = _dir(data_dir)
        sample_dir = _dir(data_dir, 'mzq_data' if control else 'mq_data')
        if derivatives[1]:
            assert control and hasattr(self.observable.problem, 'Cz')
        plan = self.initialize_sampling(derivatives, input_decoder=input_decoder, input_encoder=input_encoder,
                                        output_decoder=output_decoder, output_encoder=output_encoder)
        sinks = self._make_sinks(plan, data_dir)
        allowed_failures = 10 * n_samples + 100        # upstream tries for ever
        queued = sinks.get('Jz')

        def points(first, count, materialise):
            index = first
            while index < first + count:
                written = []
                held = len(queued._held) if queued is not None else 0
                try:
                    self.parRandom.normal(1, self.noise)
                    self.m.zero()
                    if self.settings['reset_initial_guess']:
                        self.u.zero()
                    self.prior.sample(self.noise, self.m)
                    point = [self.u, self.m, None]
                    if control:
                        self.control_distribution.sample(self.z)
                        point.append(self.z)
                    self.observable.solveFwd(self.u, point)
                    self.observable.setLinearizationPoint(point)
                    q = self.observable.evalu(self.u).get_local()
                    row = materialise(index)            # a failing incremental / adjoint solve redraws the sample as well
                    for stem, values in (('m', self.m.get_local()), ('q', q)) + ((('z', self.z.get_local()),) if control else ()):
                        written.append(sample_dir + '%s_sample_%d.npy' % (stem, index))
                        np.save(written[-1], values)
                except Exception as exc:                # noqa: BLE001 -- "issue perhaps with the forward solve, moving on"
                    self.exceptions_count += 1
                    if queued is not None:              # a save that fails AFTER materialise() returned: its queued control
                        del queued._held[held:]         # Jacobian belongs to the attempt that is being thrown away
                    for path in written:                # never leave m / q files of an index whose sample is drawn again
                        if os.path.exists(path):
                            os.remove(path)
                    if self.settings['save_failed_solves']:
                        skipped = _dir(data_dir, 'skipped')
                        np.save(skipped + 'm_sample_%d.npy' % self.exceptions_count, self.m.get_local())
                        if self.z is not None:
                            np.save(skipped + 'z_sample_%d.npy' % self.exceptions_count, self.z.get_local())
                    if self.exceptions_count > allowed_failures:
                        raise RuntimeError("DataGenerator.generate: %d failed forward solves (last: %r)"
                                           % (self.exceptions_count, exc)) from exc
                    continue
                yield row
                index += 1

        self._run_chunks(n_samples, sinks, points)
        if self.settings['verbose']:
            print("Total exceptions: %d" % self.exceptions_count)
        if compress:
            compress_dataset(data_dir, derivatives=derivatives, clean_up=clean_up, has_z_data=hasattr(self.observable.problem, 'Cz'),
                             input_decoder=input_decoder, output_decoder=output_decoder, input_encoder=input_encoder,
                             output_encoder=output_decoder if output_encoder is None else output_encoder)

    # ---- derivatives at stored samples, in a given output basis (dataGenerator.py:300-356)
    def compute_jacobians_in_subspace(self, derivatives, output_decoder, data_file_name, data_dir, output_encoder=None,
                                      compress=True, clean_up=True, compress_derivatives_only=True):       # (last one: unused upstream too)
        data_dir = _dir(data_dir)
        plan = self.initialize_sampling(derivatives, output_decoder=output_decoder, output_encoder=output_encoder)
        sinks = self._make_sinks(plan, data_dir)
        stored = np.load(data_dir + data_file_name)
        m_data, u_data = stored['m_data'], stored['q_data']
        z_data = stored['z_data'] if self.control_distribution is not None else None

        def points(first, count, materialise):
            for index in range(first, first + count):
                self.m.set_local(m_data[index])
                self.u.set_local(u_data[index])
                point = [self.u, self.m, None]
                if z_data is not None:
                    self.z.set_local(z_data[index])
                    point.append(self.z)
                self.observable.setLinearizationPoint(point)
                yield materialise(index)

        self._run_chunks(m_data.shape[0], sinks, points)
        if compress:
            has_z = hasattr(self.observable.problem, 'Cz')
            # the samples are re-archived with the derivatives when their per-sample files are still there (two_step_generate
            # leaves them), as upstream does; with only the archive on disk there is nothing to redo
            samples_on_disk = os.path.isdir(os.path.join(data_dir, 'mzq_data' if has_z else 'mq_data'))
            compress_dataset(data_dir, derivatives=derivatives, clean_up=clean_up, has_z_data=has_z,
                             output_decoder=output_decoder, output_encoder=output_decoder if output_encoder is None else output_encoder,
                             derivatives_only=not samples_on_disk)

    # ---- states first, POD of them, Jacobians in the POD basis (dataGenerator.py:251-297)
    def two_step_generate(self, n_samples, n_samples_pod=None, derivatives=(0, 0), pod_rank=None, data_dir='data/test/',
                          compress=True, clean_up=True, pod_method='hep', pod_shifted=True, M_output=None):
        from .projectors import PODProjectorFromData
        assert type(self.observable.B) is StateSpaceIdentityOperator          # a full-state problem
        n_samples_pod = n_samples if n_samples_pod is None else n_samples_pod
        assert pod_rank <= n_samples_pod, "number of samples for POD needs to be greater than rank of projector"
        data_dir = _dir(data_dir)
        self.generate(n_samples, derivatives=(0, 0), data_dir=data_dir, compress=True, clean_up=False)
        data_file_name = 'mzq_data.npz' if self.control_distribution is not None else 'mq_data.npz'
        u_data = np.load(data_dir + data_file_name)['q_data'][:n_samples_pod]
        if M_output is None:
            M_output = getattr(self.observable.B, 'M', None)
        POD = PODProjectorFromData(self.observable.problem.Vh, M_output=_as_sparse(M_output), ctx=self.ctx)
        d_POD, phi, Mphi, u_shift = POD.construct_subspace(u_data, pod_rank, shifted=pod_shifted, method=pod_method, verify=False)
        r = pod_rank - 1 if pod_shifted else pod_rank
        orth_error = np.linalg.norm(Mphi[:, :r].T @ phi[:, :r] - np.eye(r))
        if self.settings['verbose']:
            print('||Psi^*Psi - I|| = ', orth_error)
        assert orth_error < 1e-5
        pod_dir = _dir(data_dir, 'POD')
        np.save(pod_dir + 'POD_decoder.npy', phi)
        np.save(pod_dir + 'POD_encoder.npy', Mphi)
        np.save(pod_dir + 'd_POD.npy', d_POD)
        np.save(pod_dir + 'POD_shift.npy', u_shift)
        self.compute_jacobians_in_subspace(derivatives=derivatives, output_decoder=phi, output_encoder=Mphi,
                                           data_file_name=data_file_name, data_dir=data_dir, compress=compress, clean_up=clean_up)


def _as_sparse(M):
    import scipy.sparse as sp
    if M is None or sp.issparse(M):
        return M
    if hasattr(M, 'getValuesCSR'):
        row, col, val = M.getValuesCSR()
        return sp.csr_matrix((val, col, row))
    if hasattr(M, 'A'):                      # the numpy / scipy matrix a test double carries
        return sp.csr_matrix(M.A)
    return sp.csr_matrix(M)


# ------------------------------------------------------------------ per-sample files -> one archive per kind
def _stacked(folder, stem, count):
    first = np.load(folder + stem % 0 + '.npy')
    out = np.empty((count,) + first.shape)
    out[0] = first
    for i in range(1, count):
        out[i] = np.load(folder + stem % i + '.npy')
    return out


def _run_length(folder, stem):
    """How many consecutive files stem % 0, stem % 1, ... exist."""
    n = 0
    while os.path.exists(folder + stem % n + '.npy'):
        n += 1
    return n


def compress_dataset(file_path, derivatives=(0, 0), clean_up=True, has_z_data=False, input_decoder=None, output_decoder=None,
                     input_encoder=None, output_encoder=None, derivatives_only=False):
    """dataGenerator.py:495-700: gather the per-sample files under ``file_path`` into ``mq_data.npz`` / ``mzq_data.npz`` and, per
    derivative, whichever of ``J[z]starPhi_data.npz`` / ``J[z]Psi_data.npz`` / ``J[z]svd_data.npz`` is complete on disk (the bases
    go into the archive with it, under the reference's keys, when they are given); remove the per-sample folders when
    ``clean_up``.  Samples are numbered from 0 without gaps, as ``DataGenerator`` writes them."""
    file_path = os.path.join(file_path, '')
    if derivatives[1]:
        assert has_z_data
    sample_dir = file_path + ('mzq_data' if has_z_data else 'mq_data') + os.sep
    count = None
    if not derivatives_only:
        count = _run_length(sample_dir, 'm_sample_%d')
        if count == 0:
            raise FileNotFoundError("compress_dataset: no m_sample_0.npy under %s" % sample_dir)
        names = ['m', 'q'] + (['z'] if has_z_data else [])
        for name in names:
            have = _run_length(sample_dir, name + '_sample_%d')
            assert have >= count, "compress_dataset: %s_sample_%d.npy missing under %s" % (name, have, sample_dir)
        arrays = {name + '_data': _stacked(sample_dir, name + '_sample_%d', count) for name in names}
        np.savez_compressed(file_path + ('mzq_data.npz' if has_z_data else 'mq_data.npz'), **arrays)
    bases = {'adjoint_basis': dict(Phi=output_decoder, MPhi=output_encoder),
             'forward_basis': dict(Psi=input_decoder, input_encoder=input_encoder), 'svd': {}}
    for wanted, tag, stems in ((derivatives[0], 'J', _J_STEMS), (derivatives[1], 'Jz', _JZ_STEMS)):
        if not wanted:
            continue
        folder = file_path + tag + '_data' + os.sep
        found = False
        for kind, label in (('adjoint_basis', 'starPhi'), ('forward_basis', 'Psi'), ('svd', 'svd')):
            have = min(_run_length(folder, stem) for stem in stems[kind])
            if have == 0 or (count is not None and have < count):
                continue
            found = True
            n = have if count is None else count
            if kind == 'svd':
                z = 'z' if tag == 'Jz' else ''
                keys = ('U%s_data' % z, 'sigma%s_data' % z, 'V%s_data' % z)
                arrays = {key: _stacked(folder, stem, n) for key, stem in zip(keys, stems[kind])}
            else:
                arrays = {tag + label + '_data': _stacked(folder, stems[kind][0], n)}
                arrays.update({key: np.asarray(b) for key, b in bases[kind].items() if b is not None})
            np.savez_compressed(file_path + tag + label + '_data.npz', **arrays)
        assert found, "compress_dataset: no complete set of %s files under %s" % (tag, folder)
    if clean_up:
        doomed = [] if derivatives_only else [sample_dir]
        doomed += [file_path + tag + '_data' for wanted, tag in ((derivatives[0], 'J'), (derivatives[1], 'Jz')) if wanted]
        for folder in doomed:
            shutil.rmtree(folder, ignore_errors=True)
