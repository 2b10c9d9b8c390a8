"""Minimal NeuralModule / NmTensor / NeuralModuleFactory layer: the drop-in boundary.

The reference drives its modules through ``PtActions.__nm_graph_forward_pass``
(nemo/backends/pytorch/actions.py:380-442): the user wires modules by calling them with NmTensor
keyword arguments (a DAG is recorded, port names and neural types are checked,
nemo/core/neural_modules.py:423-523), then ``NeuralModuleFactory.infer(tensors=[...])``
(nemo/core/neural_factory.py:623-671 -> actions.py:639-821) topologically sorts the DAG, pulls
batches from the single DataLayerNM, moves them to the device and calls every module as
``module(force_pt=True, **{port_name: torch.Tensor})``.

This file re-implements exactly that protocol (same class / method / port / error names) in a few
hundred lines so that infer.py-style code runs unchanged on top of the HIP modules in asr.py; the
reference's ``nemo`` package itself is never imported (it does not exist on the GPU box).
Training machinery (optimizers, callbacks, amp, DDP) is out of scope.
"""
import collections
import enum
import os
import uuid
from abc import ABC, abstractmethod

import torch
import torch.nn as nn


# --------------------------------------------------------------------------- enums / errors
class DeviceType(enum.Enum):          # nemo/core/neural_factory.py:78-83
    GPU = 1
    CPU = 2
    AllGpu = 3


class Backend(enum.Enum):             # nemo/core/neural_factory.py:44-49
    PyTorch = 1
    NotSupported = 2


class ModelMode(enum.Enum):
    train = 0
    eval = 1


class NeuralPortNameMismatchError(Exception):          # neural_types/neural_type.py
    pass


class NeuralPortNmTensorMismatchError(Exception):
    pass


class NeuralTypeComparisonResult(enum.Enum):           # neural_types/comparison.py:23-34
    SAME = 0
    LESS = 1
    GREATER = 2
    DIM_INCOMPATIBLE = 3
    TRANSPOSE_SAME = 4
    INCOMPATIBLE = 6


# --------------------------------------------------------------------------- neural types
class ElementType:
    """Base of the element types used on the ASR ports (neural_types/elements.py)."""
    parents = ()

    def __init__(self, **params):
        self.params = params

    def compare(self, other):
        if type(self) is type(other):
            same = all(self.params.get(k) == other.params.get(k) or self.params.get(k) is None
                       or other.params.get(k) is None for k in set(self.params) | set(other.params))
            return NeuralTypeComparisonResult.SAME if same else NeuralTypeComparisonResult.INCOMPATIBLE
        if isinstance(other, type(self)):
            return NeuralTypeComparisonResult.GREATER
        if isinstance(self, type(other)):
            return NeuralTypeComparisonResult.LESS
        return NeuralTypeComparisonResult.INCOMPATIBLE

    def __repr__(self):
        return f"{type(self).__name__}({', '.join(f'{k}={v}' for k, v in self.params.items())})"


class VoidType(ElementType):
    def compare(self, other):
        return NeuralTypeComparisonResult.SAME


class ChannelType(ElementType):
    pass


class AudioSignal(ElementType):
    def __init__(self, freq=None):
        super().__init__(freq=freq)


class LengthsType(ElementType):
    pass


class SpectrogramType(ChannelType):
    pass


class MelSpectrogramType(SpectrogramType):
    pass


class AcousticEncodedRepresentation(ChannelType):
    pass


class LogprobsType(ElementType):
    pass


class PredictionsType(ElementType):
    pass


class NeuralType:
    """axes ('B','D','T') + element type (neural_types/neural_type.py:34-187)."""

    def __init__(self, axes=None, elements_type=None, optional=False):
        self.axes = tuple(axes) if axes is not None else None
        self.elements_type = elements_type if elements_type is not None else VoidType()
        self.optional = optional

    def compare(self, second):
        second = second.ntype if isinstance(second, NmTensor) else second
        if self.axes is None or second.axes is None:
            dims = NeuralTypeComparisonResult.SAME
        elif self.axes == second.axes:
            dims = NeuralTypeComparisonResult.SAME
        elif sorted(self.axes) == sorted(second.axes):
            dims = NeuralTypeComparisonResult.TRANSPOSE_SAME
        else:
            dims = NeuralTypeComparisonResult.DIM_INCOMPATIBLE
        if dims != NeuralTypeComparisonResult.SAME:
            return dims
        return self.elements_type.compare(second.elements_type)

    def __repr__(self):
        return f"NeuralType(axes={self.axes}, elements_type={self.elements_type!r})"


class NmTensor:
    """Symbolic edge of the DAG (neural_types/neural_type.py:190-246)."""

    def __init__(self, producer, producer_args, name, ntype=None):
        self.producer = producer
        self.producer_args = producer_args
        self.name = name
        self.ntype = ntype if ntype is not None else NeuralType()
        self._uuid = str(uuid.uuid4())

    @property
    def unique_name(self):
        return f"{self.name}~~~{self.producer.unique_instance_id if self.producer else None}~~~{self._uuid}"

    def compare(self, other):
        return self.ntype.compare(other)

    def __repr__(self):
        return f"NmTensor({self.name} <- {self.producer})"


# --------------------------------------------------------------------------- modules
class NeuralModule(ABC):
    """Port-typed module (nemo/core/neural_modules.py:60-128, 377-393, 423-523)."""

    def __init__(self):
        self._factory = NeuralModuleFactory.get_default_factory()
        self._placement = self._factory.placement if self._factory else DeviceType.GPU
        self._uuid = str(uuid.uuid4())

    @property
    @abstractmethod
    def input_ports(self):
        ...

    @property
    @abstractmethod
    def output_ports(self):
        ...

    @property
    def placement(self):
        return self._placement

    @property
    def factory(self):
        return self._factory

    @property
    def unique_instance_id(self):
        return self._uuid

    def __str__(self):
        return self.__class__.__name__      # checkpoint file names use it (callbacks.py:283-287)

    def __call__(self, **kwargs):
        in_defs, out_defs = self.input_ports, self.output_ports
        for port, value in kwargs.items():
            if port not in in_defs:
                raise NeuralPortNameMismatchError(f"Wrong input port name: {port}")
            res = in_defs[port].compare(value)
            if res not in (NeuralTypeComparisonResult.SAME, NeuralTypeComparisonResult.GREATER):
                raise NeuralPortNmTensorMismatchError(
                    f"\n\nIn {type(self).__name__}. \nPort: {port} and a NmTensor it was fed are \n"
                    f"of incompatible neural types:\n\n{in_defs[port]} \n\n and \n\n{value}"
                    f"\n\nType comparison result: {res}")
        outs = [NmTensor(producer=self, producer_args=kwargs, name=n, ntype=t) for n, t in out_defs.items()]
        if len(outs) == 1:
            return outs[0]
        return collections.namedtuple(f"{type(self).__name__}Output", list(out_defs))(*outs)


def get_cuda_device(placement):
    """nemo/utils/helpers.py:94-104."""
    return torch.device("cuda") if placement in (DeviceType.GPU, DeviceType.AllGpu) else torch.device("cpu")


class TrainableNM(NeuralModule, nn.Module):
    """NeuralModule that is also an nn.Module (nemo/backends/pytorch/nm.py:13-129)."""

    def __init__(self):
        NeuralModule.__init__(self)
        nn.Module.__init__(self)
        self._device = get_cuda_device(self.placement)

    def __call__(self, *inputs, force_pt=False, **kwargs):
        if inputs or force_pt:
            return nn.Module.__call__(self, *inputs, **kwargs)
        return NeuralModule.__call__(self, **kwargs)

    def get_weights(self):
        return {n: (p, p.requires_grad) for n, p in self.named_parameters()}

    def set_weights(self, name2weight, name2name_and_transform=None):
        if name2name_and_transform is not None:
            raise NotImplementedError("Transforms are not currently supported for set_weights")
        if name2weight:
            self.load_state_dict({k: v[0] for k, v in name2weight.items()})

    def tie_weights_with(self, module, weight_names, name2name_and_transform=None):
        raise NotImplementedError("weight tying is a training feature; not part of the inference path")

    def save_to(self, path):
        torch.save(self.state_dict(), path)

    def restore_from(self, path, local_rank=0):
        # nm.py:97-103; weights are staged on the host and re-packed for the HIP kernels on next forward
        self.load_state_dict(torch.load(path, map_location="cpu"))

    def freeze(self, weights=None):
        for n, p in self.named_parameters():
            if weights is None or n in weights:
                p.requires_grad = False

    def unfreeze(self, weights=None):
        for n, p in self.named_parameters():
            if weights is None or n in weights:
                p.requires_grad = True

    @property
    def num_weights(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)


class NonTrainableNM(NeuralModule):
    """nemo/backends/pytorch/nm.py:132-184."""

    def __init__(self):
        NeuralModule.__init__(self)
        self._device = get_cuda_device(self.placement)

    def __call__(self, force_pt=False, *inputs, **kwargs):
        if inputs or force_pt:
            with torch.no_grad():
                return self.forward(*inputs, **kwargs)
        return NeuralModule.__call__(self, **kwargs)

    def forward(self, *a, **k):
        raise NotImplementedError

    def get_weights(self):
        return None

    def set_weights(self, *a, **k):
        return None

    def save_to(self, path):
        return None

    def restore_from(self, path):
        return None

    @property
    def num_weights(self):
        return 0


class DataLayerNM(NeuralModule):
    """Source of the DAG (nemo/backends/pytorch/nm.py:187-320): exposes ``dataset`` or ``data_iterator``."""

    def __init__(self):
        NeuralModule.__init__(self)
        self._device = get_cuda_device(self.placement)
        self._batch_size = 1
        self._num_workers = 0
        self._shuffle = False

    @property
    def input_ports(self):
        return {}

    @abstractmethod
    def __len__(self):
        ...

    @property
    @abstractmethod
    def dataset(self):
        ...

    @property
    @abstractmethod
    def data_iterator(self):
        ...

    batch_size = property(lambda self: self._batch_size)
    shuffle = property(lambda self: self._shuffle)
    num_workers = property(lambda self: self._num_workers)


# --------------------------------------------------------------------------- executor + factory
def _pad_to(t, shape):
    out = t.new_zeros(tuple(int(s) for s in shape))
    out[tuple(slice(0, s) for s in t.shape)] = t
    return out


_GATHER_DTYPES = [torch.float32, torch.int64, torch.int32, torch.float64, torch.float16, torch.bfloat16, torch.int16,
                  torch.int8, torch.uint8, torch.bool]


def _gather_ragged(v, device):
    """all_gather of one port value per rank, as PtActions._infer does it for tensors (actions.py:774-807:
    all_gather(shape) -> pad to max -> all_gather(padded) -> de-pad), extended by two cases: ``v is None`` = this rank
    has no batch left (its empty part is dropped), and a non-tensor value (the batched beam decoder's list of
    transcripts), which travels through all_gather_object.  EVERY rank calls this for EVERY port on EVERY step and all
    decisions are taken from the descriptors all ranks hold after the first collective -- a rank may not decide from its
    own value alone whether the port is gathered (round 2 did: a rank that still had a list skipped the call, a rank
    that had run out entered it, and its all_gathers paired up with the other rank's next port).
    Returns the parts of the ranks that had one, in rank order; [] when no rank had anything."""
    import torch.distributed as dist
    world = dist.get_world_size()
    desc = torch.full((10,), -1, dtype=torch.int64, device=device)       # [ndim | -1 nothing | -2 object, dtype code, dims...]
    is_tensor = isinstance(v, torch.Tensor)
    if is_tensor:
        if v.dim() > 8:
            raise ValueError("tensors of more than 8 dimensions are not gathered")
        desc[0], desc[1] = v.dim(), _GATHER_DTYPES.index(v.dtype)
        if v.dim():
            desc[2 : 2 + v.dim()] = torch.tensor(v.shape, dtype=torch.int64)
    elif v is not None:
        desc[0] = -2
    descs = [torch.empty_like(desc) for _ in range(world)]
    dist.all_gather(descs, desc)
    descs = [d.tolist() for d in descs]
    have = [d for d in descs if d[0] >= 0]
    if any(d[0] == -2 for d in descs):
        if have:
            raise ValueError("ranks returned a tensor and a non-tensor value for one port")
        objs = [None] * world
        dist.all_gather_object(objs, v)
        return [o for o, d in zip(objs, descs) if d[0] == -2]
    if not have:
        return []
    ndim, dtype = have[0][0], _GATHER_DTYPES[have[0][1]]
    if any(d[0] != ndim or d[1] != have[0][1] for d in have):
        raise ValueError("ranks returned tensors of different rank / dtype for one port")
    mx = [max(d[2 + i] for d in have) for i in range(ndim)]
    padded = _pad_to(v, mx) if is_tensor else torch.zeros(mx, dtype=dtype, device=device)
    gathered = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(gathered, padded)
    return [g[tuple(slice(0, int(n)) for n in d[2 : 2 + ndim])] for g, d in zip(gathered, descs) if d[0] >= 0]


class _Actions:
    """The inference half of PtActions (actions.py:104-221, 380-442, 639-821)."""

    def __init__(self, factory):
        self.factory = factory

    @staticmethod
    def topo_sort(tensors):
        """DFS from the requested tensors back to the data layer; returns modules in call order with the
        NmTensor kwargs each was wired with and its output NmTensors."""
        order, seen = [], {}

        def visit(t):
            m = t.producer
            if m is None:
                return
            key = (m.unique_instance_id, id(t.producer_args))
            if key in seen:
                seen[key][2][t.name] = t
                return
            for a in (t.producer_args or {}).values():
                visit(a)
            entry = (m, t.producer_args or {}, {t.name: t})
            seen[key] = entry
            order.append(entry)

        for t in tensors:
            visit(t)
        dls = [e for e in order if isinstance(e[0], DataLayerNM)]
        if len(dls) != 1 or order[0] is not dls[0]:
            raise ValueError("The DAG must have exactly one DataLayerNM and it must be the first module "
                             "(actions.py:193-198)")
        return order

    def forward_pass(self, chain, registered):
        """__nm_graph_forward_pass in eval mode (actions.py:380-442)."""
        for module, call_args, outs in chain[1:]:
            if isinstance(module, nn.Module):
                module.eval()                     # NOT reached for NonTrainableNM (quirk Q1)
            call_set = {port: registered[t.unique_name] for port, t in call_args.items()}
            result = module(force_pt=True, **call_set)
            # A module with ONE output port hands back that port's value, whatever its type: the batched beam decoder
            # returns a Python list of B transcripts, which must not be zipped against its single port (the reference
            # only ever sees a str there, batch 1: beam_search_decoder.py:95-102).
            if len(module.output_ports) == 1 and not isinstance(result, tuple):
                result = (result,)
            elif not isinstance(result, (tuple, list)):
                result = (result,)
            # results are zipped against output_ports order (actions.py:430-442)
            for name, value in zip(module.output_ports, result):
                if name in outs:
                    registered[outs[name].unique_name] = value

    def infer(self, tensors, verbose=False, offload_to_cpu=True):
        chain = self.topo_sort(tensors)
        dl = chain[0][0]
        dl_out_names = list(dl.output_ports)
        # NmTensors the data layer produced, by port name (any consumer's kwargs or requested tensors)
        dl_tensors = {}
        for _, args, outs in chain:
            for t in list(args.values()) + list(outs.values()):
                if t.producer is dl:
                    dl_tensors[t.name] = t
        import torch.distributed as dist
        distributed = dl.placement == DeviceType.AllGpu
        if distributed:
            assert dist.is_initialized(), "AllGpu placement needs an initialised process group"
        if dl.dataset is not None:
            sampler = None
            if distributed:
                sampler = torch.utils.data.distributed.DistributedSampler(dataset=dl.dataset, shuffle=dl.shuffle)
                sampler.set_epoch(0)
            loader = torch.utils.data.DataLoader(dataset=dl.dataset, sampler=sampler, batch_size=dl.batch_size,
                                                 shuffle=False if sampler is not None else dl.shuffle,
                                                 num_workers=dl.num_workers,
                                                 collate_fn=getattr(dl, "collate_fn", None))
        else:
            loader = dl.data_iterator
        values = {t.unique_name: [] for t in tensors}
        n_steps = None
        if distributed:
            # The reference pads every rank to the same number of batches (DistributedSampler); a data layer that shards
            # by itself (data_iterator) may leave ranks with different counts -- 5 utterances, 2 ranks, batch 2 gives 2
            # and 1 -- and the rank with more would wait in all_gather for ever.  Every rank therefore takes
            # max-over-ranks steps; a rank that has run out joins the collectives with an empty contribution.
            # A loader without a length (the reference accepts a plain iterator / generator as data_iterator,
            # actions.py:697-707) agrees on termination step by step instead: MAX over ranks of "I still have a batch".
            n = torch.tensor([len(loader) if hasattr(loader, "__len__") else -1], dtype=torch.int64, device=dl._device)
            lo = n.clone()
            dist.all_reduce(n, op=dist.ReduceOp.MAX)
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            n_steps = int(n.item()) if int(lo.item()) >= 0 else -1      # -1: some rank cannot tell -> per-step flag
        with torch.no_grad():
            it = iter(loader)
            step = 0
            while True:
                data = next(it, None)
                if n_steps == -1:
                    more = torch.tensor([0 if data is None else 1], dtype=torch.int64, device=dl._device)
                    dist.all_reduce(more, op=dist.ReduceOp.MAX)
                    if int(more.item()) == 0:
                        break
                elif data is None and (n_steps is None or step >= n_steps):
                    break
                step += 1
                registered = None
                if data is not None:
                    if isinstance(data, torch.Tensor):
                        data = (data,)
                    batch = [d.to(dl._device) if isinstance(d, torch.Tensor) else d for d in data]   # H2D: actions.py:740-746
                    registered = {dl_tensors[n].unique_name: v for n, v in zip(dl_out_names, batch) if n in dl_tensors}
                    self.forward_pass(chain, registered)
                for t in tensors:
                    v = registered[t.unique_name] if registered is not None else None
                    if distributed:
                        parts = _gather_ragged(v, dl._device)
                        if offload_to_cpu:
                            parts = [p.cpu() if isinstance(p, torch.Tensor) else p for p in parts]
                        if dist.get_rank() == 0:
                            values[t.unique_name] += parts
                    elif v is not None:
                        if offload_to_cpu and isinstance(v, torch.Tensor):
                            v = v.cpu()
                        values[t.unique_name].append(v)
        if distributed and dist.get_rank() != 0:
            return None
        return [values[t.unique_name] for t in tensors]


class NeuralModuleFactory:
    """nemo/core/neural_factory.py:251-415, 623-671 (inference subset)."""

    _DEFAULT = None

    def __init__(self, backend=Backend.PyTorch, local_rank=None, placement=None, random_seed=None, **_ignored):
        self._local_rank = local_rank
        self._world_size = 1
        if placement is None:
            placement = DeviceType.AllGpu if local_rank is not None else DeviceType.GPU
        self._placement = placement
        if backend != Backend.PyTorch:
            raise NotImplementedError("Only Pytorch backend is currently supported.")
        if placement in (DeviceType.GPU, DeviceType.AllGpu) and not torch.cuda.is_available():
            raise ValueError("You requested to use GPUs but CUDA is not installed. You can try running using"
                             " CPU-only. To do this, instantiate your factory with placement=DeviceType.CPU")
        if random_seed is not None:
            torch.manual_seed(random_seed)
        if local_rank is not None:
            import torch.distributed as dist
            torch.cuda.set_device(local_rank)
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                dist.init_process_group(backend="nccl", init_method="env://")   # neural_factory.py:342-343 (RCCL)
            self._world_size = dist.get_world_size()
        self._trainer = _Actions(self)
        NeuralModuleFactory._DEFAULT = self

    @classmethod
    def get_default_factory(cls):
        return cls._DEFAULT

    @classmethod
    def set_default_factory(cls, factory):
        cls._DEFAULT = factory

    @classmethod
    def reset_default_factory(cls):
        cls._DEFAULT = None

    placement = property(lambda self: self._placement)
    world_size = property(lambda self: self._world_size)
    local_rank = property(lambda self: self._local_rank)

    def infer(self, tensors, checkpoint_dir=None, ckpt_pattern="", verbose=True, cache=False, use_cache=False,
              offload_to_cpu=True, modules_to_restore=None):
        if cache or use_cache:
            raise NotImplementedError("DAG caching is not implemented")
        return self._trainer.infer(tensors, verbose=verbose, offload_to_cpu=offload_to_cpu)
