"""Flat parameter / gradient buffers, the fused Adam step and the data-parallel gradient exchange.

Replaces ``th.optim.Adam(learnable_params, lr)`` (reference models/learner.py:194-199,495) with ONE kernel over one
contiguous buffer, and adds what the reference does not have: one RCCL all-reduce (sum) of the flat gradient bucket
per step across the GPUs of a node.  Parameters stay ``nn.Parameter`` objects (views into the flat
buffer) so ``state_dict()``, ``named_parameters()`` and LossManager.reg_params keep working.
"""
import torch
import torch.distributed as dist

from . import ops


class FlatParams(object):
    """Re-homes every parameter of `module` into one contiguous fp32 buffer (+ a matching gradient buffer)."""

    ALIGN = 4  # floats (16 bytes)
    TAIL = 16  # loss scalars riding behind the gradients (slot 0 = total loss: a NaN / inf on ANY rank shows on EVERY rank)

    def __init__(self, module):
        params = [p for p in module.parameters() if p.requires_grad]
        if not params:
            raise ValueError("module has no trainable parameter")
        device = params[0].device
        offsets, total = [], 0
        for p in params:
            offsets.append(total)
            total += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.flat = torch.zeros(total, dtype=torch.float32, device=device)
        # the gradient BUCKET = every parameter's gradient + a short tail of per-step scalars (total loss, the individual
        # losses): whatever is exchanged between the GPUs of a node travels in this one buffer, one collective per step
        self.bucket = torch.zeros(total + self.TAIL, dtype=torch.float32, device=device)
        self.grad = self.bucket[:total]
        self.tail = self.bucket[total:]
        self.params, self.offsets = params, offsets
        with torch.no_grad():
            for i, (p, off) in enumerate(zip(params, offsets)):
                n = p.numel()
                self.flat[off:off + n].copy_(p.detach().reshape(-1))
                p.data = self.flat[off:off + n].view(p.shape)
                p.grad = self.grad[off:off + n].view(p.shape)
                p._srlz_flat = (self, i)  # lets srlz.ops write gradient contributions into the staging buckets
        # -- direct gradient delivery -------------------------------------------------------------------------------
        # autograd would run one `grad += new` kernel per parameter and contribution (two frames -> ~70 tiny launches
        # per step).  Instead the ops' backward functions ask grad_buffer() where to let their kernels write a
        # parameter's gradient (k-th request of a pass -> stage k, a copy of the bucket at a FIXED address), return None
        # to autograd, and deliver() folds the stages into the bucket with one launch that also clears them.
        self.stage = torch.zeros((self.NSTAGE, total), dtype=torch.float32, device=device)
        self._served = [0] * len(params)
        self._dirty = False

    NSTAGE = 3  # two frames + one regulariser contribution per parameter

    def grad_buffer(self, index):
        """Where the next gradient contribution of parameter `index` should be written, or None (all stages used in this
        pass: the caller allocates and returns the gradient to autograd as usual)."""
        k = self._served[index]
        if k >= self.NSTAGE:
            return None
        self._served[index] = k + 1
        self._dirty = True
        p, off = self.params[index], self.offsets[index]
        return self.stage[k, off:off + p.numel()].view(p.shape)

    def deliver(self):
        """Fold the staged contributions into the flat gradient (idempotent; call after backward)."""
        if self._dirty:
            ops.fold_grads(self.grad, self.stage)
            self._dirty = False

    def discard(self):
        """Drop the staged contributions of a backward pass whose gradients are not used (validation minibatch)."""
        if self._dirty:
            self.stage.zero_()
            self._dirty = False

    def put_scalars(self, scalars):
        """Per-step scalars (0-dim device tensors; total loss first) into the bucket tail, with one launch."""
        if len(scalars) > self.TAIL:
            raise ValueError("at most %d scalars ride in the gradient bucket (got %d)" % (self.TAIL, len(scalars)))
        torch.stack([s.detach().reshape(()) for s in scalars], out=self.tail[:len(scalars)])

    def read_scalars(self, count):
        """The first `count` tail slots as Python floats, averaged over the ranks (they were summed by the collective) —
        the step's ONE device->host copy.  Identical on every rank after allreduce_gradients / allreduce_scalars."""
        _, size = world()
        values = self.tail[:count].tolist()
        return values if size == 1 else [v / size for v in values]

    def read_scalars_async(self, count):
        """read_scalars without the wait: the copy of the first `count` tail slots into pinned host memory is queued behind the step
        that wrote them and a ticket comes back; `scalars(ticket)` waits for THAT copy only.  learn() redeems a step's ticket after it
        has launched the next step, so the host runs one step ahead and the GPU never idles while ~100 launches are being issued
        (a blocking read per step cost 0.4-0.8 ms of a 2.4 / 13 ms step).  Two host buffers alternate: at most one ticket is
        outstanding when the next one is taken."""
        if not self.tail.is_cuda:
            return (self.tail[:count].clone(), count, None)
        if getattr(self, "_host", None) is None:
            self._host = [torch.empty(self.TAIL, dtype=torch.float32).pin_memory() for _ in range(2)]
            self._host_i = 0
        buf = self._host[self._host_i]
        self._host_i ^= 1
        buf[:count].copy_(self.tail[:count], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.tail.device))
        return (buf, count, ev)

    def scalars(self, ticket):
        """The values of a read_scalars_async ticket as Python floats (mean over the ranks, like read_scalars)."""
        buf, count, ev = ticket
        if ev is not None:
            ev.synchronize()
        _, size = world()
        values = buf[:count].tolist()
        return values if size == 1 else [v / size for v in values]

    def zero_grad(self):
        self.grad.zero_()
        if self._dirty:  # a backward pass whose gradients were neither delivered nor discarded
            self.stage.zero_()
            self._dirty = False
        self._served = [0] * len(self.params)
        # autograd accumulates in place into an existing .grad, so the views persist; re-attach if someone dropped them
        for p, off in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                p.grad = self.grad[off:off + p.numel()].view(p.shape)


class FusedAdam(object):
    """torch.optim.Adam semantics (betas (0.9, 0.999), eps 1e-8, no weight decay / amsgrad) over a FlatParams.

    Parameters that never receive a gradient keep a zero gradient, zero moments and therefore a zero update — the
    same end state as torch skipping ``grad is None`` parameters.
    """

    def __init__(self, flat_params, lr, betas=(0.9, 0.999), eps=1e-8):
        self.fp = flat_params
        self.lr, self.betas, self.eps = lr, betas, eps
        self.m = torch.zeros_like(flat_params.flat)
        self.v = torch.zeros_like(flat_params.flat)
        self.t = 0
        # device-side step counter (+ 2 floats of bias corrections): the form a captured hipGraph can replay
        self.device_step = False
        self.t_dev = torch.zeros(1, dtype=torch.int32, device=flat_params.flat.device)
        self.bc_dev = torch.zeros(2, dtype=torch.float32, device=flat_params.flat.device)

    def zero_grad(self):
        self.fp.zero_grad()

    def use_device_step(self):
        """Switch to the device-side step counter (continuing from the current step)."""
        if not self.device_step:
            self.t_dev.fill_(self.t)
            self.device_step = True

    def step(self, grad_scale=1.0):
        self.fp.deliver()
        self.t += 1  # (host mirror; in device_step mode a graph replay advances only t_dev — see steps())
        if self.device_step:
            ops.adam_step_dev(self.fp.flat, self.fp.grad, self.m, self.v, self.lr, self.t_dev, self.bc_dev, grad_scale,
                              self.betas, self.eps)
        else:
            ops.adam_step(self.fp.flat, self.fp.grad, self.m, self.v, self.lr, self.t, grad_scale, self.betas, self.eps)

    def steps(self):
        """Number of updates applied so far."""
        return int(self.t_dev.item()) if self.device_step else self.t


# ---- which collective moves the bucket -------------------------------------------------------------------------------------
# Default: torch.distributed's all_reduce (backend "nccl" = RCCL), because the launcher, the rendezvous and the process group
# are torch.distributed's anyway.  SRLZ_COMM=rccl routes the same in-place sum through the library's own C-ABI entry point
# (srlz_comm_allreduce_f32, include/srlz.h) on a communicator created from a unique id that rank 0 shares through the
# process group's store — what a host without torch.distributed would bind.
_native_ready = False


def native_comm_requested():
    import os
    return os.environ.get("SRLZ_COMM", "torch").lower() == "rccl"


def init_native_comm():
    """Create the process's RCCL communicator through the C ABI (idempotent).  Call after init_process_group (world > 1)
    with the rank's GPU current."""
    global _native_ready
    if _native_ready:
        return
    import ctypes
    from . import _cabi as C
    rank, size = world()
    nbytes = C.comm_unique_id_bytes()
    buf = ctypes.create_string_buffer(nbytes)
    if rank == 0:
        C.comm_unique_id(buf)
    payload = share_from_rank0(buf.raw)
    C.comm_init(ctypes.create_string_buffer(payload, nbytes), rank, size)
    if C.comm_world() != size:  # (the communicator's own count)
        raise RuntimeError("srlz_comm_init: the RCCL communicator has %d ranks, the process group %d" % (C.comm_world(), size))
    _native_ready = True


def destroy_native_comm():
    global _native_ready
    if _native_ready:
        from . import _cabi as C
        C.comm_destroy()
        _native_ready = False


def dist_backend():
    """Backend of the process group the ranks are launched with: "nccl" (= RCCL over xGMI, one GPU per rank — the product
    configuration) unless SRLZ_DIST_BACKEND says otherwise.  "gloo" exists for ONE purpose: several ranks sharing a GPU on a
    box with fewer GPUs than ranks (RCCL refuses two ranks on one device), so that the whole multi-rank control flow of
    learn() / train.py / bench.py can be exercised on a 1-GPU machine; the bucket then bounces through host memory."""
    import os
    return os.environ.get("SRLZ_DIST_BACKEND", "nccl").lower()


def local_device_index():
    """GPU of this rank: LOCAL_RANK, wrapped around the visible devices (ranks share GPUs only in the gloo debug topology)."""
    import os
    n = max(torch.cuda.device_count(), 1)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if local_rank >= n and dist_backend() == "nccl":
        raise RuntimeError("LOCAL_RANK %d but only %d GPU(s) visible: RCCL needs one GPU per rank (SRLZ_DIST_BACKEND=gloo "
                           "lets ranks share a GPU for functional tests)" % (local_rank, n))
    return local_rank % n


def device_identity(index=None):
    """(host name, PCI address "dddd:bb:dd.0") of the rank's GPU — what tells two ranks on one device apart from two ranks on two."""
    import socket
    if not torch.cuda.is_available():
        return socket.gethostname(), "cpu"  # (the CPU tests of the multi-rank control flow)
    index = torch.cuda.current_device() if index is None else index
    p = torch.cuda.get_device_properties(index)
    pci = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    # (partitioned GPUs can share a PCI address: the device's UUID, when the runtime reports one, tells the partitions apart)
    uuid = str(getattr(p, "uuid", "") or "")
    return socket.gethostname(), pci + ("/" + uuid if uuid.strip("0-") else "")


def rank_devices():
    """COLLECTIVE, right after init_process_group: every rank's (host, PCI address), gathered over the process group — the
    communicator's own view of the job, not the environment's.  Under backend "nccl" (RCCL: the product configuration) two ranks on one
    device, or a communicator whose size differs from WORLD_SIZE, stop EVERY rank here with a message instead of a hang inside the
    first collective (RCCL refuses the second rank of a device only after the others are already waiting for it).  The gloo debug
    topology shares devices on purpose.  Returns {"ranks": W, "devices": distinct devices, "hosts": distinct hosts, "table": [...]}."""
    import os
    rank, size = world()
    if size == 1:
        host, pci = device_identity()
        return {"ranks": 1, "devices": 1, "hosts": 1, "table": [[host, pci]]}
    table = [None] * size
    try:
        mine = list(device_identity())
    except Exception as e:  # an identity the runtime cannot report is never mistaken for a shared device
        mine = ["rank%d" % rank, "unknown (%s)" % type(e).__name__]
    dist.all_gather_object(table, mine)
    distinct = len(set(tuple(t) for t in table))
    info = {"ranks": size, "devices": distinct, "hosts": len(set(t[0] for t in table)), "table": table}
    env_world = int(os.environ.get("WORLD_SIZE", size))
    if dist.get_backend() == "nccl" and (distinct != size or env_world != size):
        shared = sorted(set(tuple(t) for t in table if table.count(t) > 1))
        raise RuntimeError("rank %d: %d ranks (WORLD_SIZE=%d) on %d distinct GPU(s); RCCL needs one GPU per rank.  Shared: %s.  "
                           "Check LOCAL_RANK / HIP_VISIBLE_DEVICES of the launcher (SRLZ_DIST_BACKEND=gloo lets ranks share a GPU "
                           "for functional tests)" % (rank, size, env_world, distinct, shared))
    return info


# ---- host cores: the loaders of all local ranks share one machine -----------------------------------------------------------
def usable_cores():
    """CPUs this process may use: min(affinity mask, cgroup cpu.max quota)."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (IOError, OSError, ValueError):
        pass
    return n


def loader_workers(requested, passes=1, local_ranks=None, cores=None):
    """Decoding threads per loader process such that  threads x passes x local ranks <= usable cores  (passes = 2 while a rank's fill
    pass decodes its dataset slice beside the first epoch).  The reference's N_WORKERS = 4 is a single-process setting; eight ranks x
    two loader processes x 4 threads on a 32-core host would oversubscribe the cores the step's own launch thread needs."""
    import os
    if local_ranks is None:
        local_ranks = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    cores = usable_cores() if cores is None else cores
    per_rank = max(1, (cores - local_ranks) // max(1, local_ranks))  # (one core per rank stays with the training process)
    return int(max(1, min(int(requested), per_rank // max(1, passes))))


def numa_cpus_of_device(index=None, sysfs="/sys"):
    """CPUs of the NUMA node the rank's GPU hangs off, intersected with the CPUs this process may use (None: unknown, one node, or an
    empty intersection — leave the affinity alone).  Read from <sysfs>/bus/pci/devices/<address>/numa_node and
    <sysfs>/devices/system/node/node<k>/cpulist.  The loader processes pin themselves to it (DataLoader(cpu_affinity=...)): decoded
    frames are written into shared memory next to the GPU's PCIe root and are not bounced between sockets."""
    import os
    try:
        pci = device_identity(index)[1].split("/")[0] if not isinstance(index, str) else index
        node = int(open(os.path.join(sysfs, "bus/pci/devices", pci, "numa_node")).read().strip())
        if node < 0:
            return None
        text = open(os.path.join(sysfs, "devices/system/node/node%d/cpulist" % node)).read().strip()
    except (IOError, OSError, ValueError, RuntimeError, AssertionError):
        return None
    cpus = set()
    for part in text.split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    allowed = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else cpus
    cpus &= set(allowed)
    return sorted(cpus) if cpus and len(cpus) < len(allowed) else None


# bench.py --gpus N: HIP-event pairs around every exchange of the instrumented steps (on the launch stream: torch.distributed's
# collective makes the current stream wait for RCCL's, so the second event completes behind the all-reduce)
_comm_events = None


def comm_timing(on):
    """Start (True) / stop (False) recording one HIP-event pair per collective of the data path."""
    global _comm_events
    _comm_events = [] if on else None


def comm_timing_report():
    """[(microseconds, bytes)] of the collectives recorded since comm_timing(True) (synchronises)."""
    if not _comm_events:
        return []
    torch.cuda.synchronize()
    return [(1e3 * a.elapsed_time(b), nbytes) for a, b, nbytes in _comm_events]


def _sum_across_ranks(t):
    if _comm_events is not None and t.is_cuda:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(torch.cuda.current_stream(t.device))
        _sum_across_ranks_impl(t)
        b.record(torch.cuda.current_stream(t.device))
        _comm_events.append((a, b, t.numel() * t.element_size()))
        return
    _sum_across_ranks_impl(t)


def _sum_across_ranks_impl(t):
    if _native_ready:
        from . import _cabi as C
        C.comm_allreduce_f32(C.ptr(t), t.numel(), C.stream())
    elif t.is_cuda and dist.get_backend() == "gloo":
        host = t.detach().cpu()  # (debug topology only, see dist_backend)
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def allreduce_gradients(flat_params):
    """One all-reduce (sum) of the whole bucket — gradients AND the scalar tail (FlatParams.put_scalars), so a NaN loss on
    one rank reaches every rank with the gradients (reference models/learner.py:520-522: NaN -> exit code 11; every rank must
    take that exit together, or the others hang in their next collective).  Returns the scale Adam applies (1/world_size)."""
    flat_params.deliver()
    rank, size = world()
    if size == 1:
        return 1.0
    _sum_across_ranks(flat_params.bucket)
    return 1.0 / size


def allreduce_scalars(flat_params):
    """Validation minibatches take no optimiser step: only the scalar tail is exchanged (ranks validate in lock-step,
    preprocessing/data_loader.py::shardOrder)."""
    _, size = world()
    if size > 1:
        _sum_across_ranks(flat_params.tail)


def average_running_stats(state_dict):
    """BatchNorm running statistics averaged over the ranks IN `state_dict` (a copy about to be checkpointed): batch
    statistics are rank-local during training (the reference's per-call semantics), the saved model must not depend on which
    rank wrote it.  One small all-reduce per checkpoint; every rank must call it."""
    _, size = world()
    keys = [k for k, v in state_dict.items() if "running_" in k and v.is_floating_point()]
    if size == 1 or not keys:
        return state_dict
    flat = torch.cat([state_dict[k].reshape(-1).float() for k in keys])
    if flat.is_cuda and dist.get_backend() == "gloo":
        flat = flat.cpu()
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat = flat.to(state_dict[keys[0]].device)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= size
    off = 0
    for k in keys:
        n = state_dict[k].numel()
        state_dict[k] = flat[off:off + n].view(state_dict[k].shape).to(state_dict[k].dtype).clone()
        off += n
    return state_dict


def all_ranks(flag):
    """True only when `flag` is true on EVERY rank (one small all-reduce, MIN; every rank must call it at the same point of the
    program).  For decisions that create or skip later collectives — e.g. whether a rank-local memory budget allows the resident
    dataset: a store on some ranks and none on others would pair one rank's slice exchange with another's gradient all-reduce."""
    _, size = world()
    if size == 1:
        return bool(flag)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    ok = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    return int(ok.item()) == 1


def share_from_rank0(obj):
    """`obj` as rank 0 computed it, on every rank (log-folder names carry a wall-clock timestamp)."""
    _, size = world()
    if size == 1:
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=0)
    return box[0]
