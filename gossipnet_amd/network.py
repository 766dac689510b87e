"""Gnet -- host-side mirror of the reference's nms_net/network.py:121-322 class `Gnet`.

Same constructor, same input names (`get_batch_spec`), same output attribute names
(`prediction`, `labels`, `weights`, `det_gt_matching`, `det_anno_iou`, `loss`, `loss_normed`,
`loss_unnormed`, `pw_feats`, `block_feats`, `neighbor_pair_idxs`, `num_dets`, `class_weights`,
`trainable_variables`), and hyper-parameters taken from the global `cfg` at construction
(reference: nms_net/config.py).  There is no TF graph: `net.run(batch)` plays the role of
`sess.run(...)` and fills the attributes.  All arithmetic happens in libgossipnet_hip.so
(hand-written HIP for gfx950) through the C ABI of include/gossipnet_hip.h; PyTorch only owns
device memory and streams.  There is no CPU fallback.
"""
import ctypes as C
import os
import math
import weakref

import numpy as np
import torch

from . import _lib
from .config import cfg


def _vp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def imfeat_param_spec(imfeat_channels):
    """Variables of the image-feature start features (network.py:223-240), appended behind the core parameters:
    flatten(roifeats) [N, crop_h*crop_w*C] -> fully_connected (-> imfeat_dim, only if > 0) -> fully_connected (-> 128).
    TF names the first layer of the scope `fully_connected`, a second one `fully_connected_1`."""
    g = cfg.gnet
    d_in = int(cfg.imfeat_crop_height) * int(cfg.imfeat_crop_width) * int(imfeat_channels)
    spec, names = [], ["gnet/reduce_imfeats/fully_connected/", "gnet/reduce_imfeats/fully_connected_1/"]
    if g.imfeat_dim > 0:
        spec += [(names[0] + "weights", (d_in, int(g.imfeat_dim))), (names[0] + "biases", (int(g.imfeat_dim),))]
        d_in = int(g.imfeat_dim)
        names = names[1:]
    spec += [(names[0] + "weights", (d_in, g.shortcut_dim)), (names[0] + "biases", (g.shortcut_dim,))]
    return spec


def param_spec(num_classes, num_blocks):
    """TF variable names + shapes in flat-buffer order (include/gossipnet_hip.h)."""
    cp = num_classes if num_classes > 1 else 1
    d = 2 * cp + 7
    g = cfg.gnet
    spec = []
    # _pw_feats_fc (network.py:324-342): num_pwfeat_fc - 1 layers of pwfeat_dim, then one of pwfeat_narrow_dim; with
    # num_pwfeat_fc = 0 (the reference's default, network.py:217-221) the blocks read the raw feature columns
    nfc = int(g.num_pwfeat_fc)
    dims = [d] + [g.pwfeat_dim] * max(nfc - 1, 0) + ([g.pwfeat_narrow_dim] if nfc > 0 else [])
    for i in range(nfc):
        spec.append(("gnet/pw_feats/fc%d/weights" % (i + 1), (dims[i], dims[i + 1])))
        spec.append(("gnet/pw_feats/fc%d/biases" % (i + 1), (dims[i + 1],)))
    pw_in = dims[-1]
    for b in range(1, num_blocks + 1):
        p = "gnet/block%d/" % b
        spec += [
            (p + "reduce_dim/weights", (g.shortcut_dim, g.reduced_dim)), (p + "reduce_dim/biases", (g.reduced_dim,)),
            (p + "pw_fc1/weights", (pw_in + 2 * g.reduced_dim, g.pairfeat_dim)),
            (p + "pw_fc1/biases", (g.pairfeat_dim,)),
            (p + "pw_fc2/weights", (g.pairfeat_dim, g.pairfeat_dim)), (p + "pw_fc2/biases", (g.pairfeat_dim,)),
            (p + "fc1/weights", (g.pairfeat_dim, g.pairfeat_dim)), (p + "fc1/biases", (g.pairfeat_dim,)),
            (p + "fc2/weights", (g.pairfeat_dim, g.shortcut_dim)), (p + "fc2/biases", (g.shortcut_dim,)),
        ]
        if g.neighbor_feats:       # network.py:356-365 (flat-buffer position: behind fc2 of the block)
            spec += [(p + "reduce_dim_neighbor/weights", (g.shortcut_dim, g.reduced_dim)),
                     (p + "reduce_dim_neighbor/biases", (g.reduced_dim,))]
    for i in (1, 2):
        p = "gnet/predict/fc%d/fully_connected/" % i
        spec += [(p + "weights", (g.predict_fc_dim, g.predict_fc_dim)), (p + "biases", (g.predict_fc_dim,))]
    p = "gnet/predict/logits/fully_connected/"
    spec += [(p + "weights", (g.predict_fc_dim, 1)), (p + "biases", (1,))]
    return spec


class DeviceBatch(object):
    """One or more images concatenated and resident on the device (block-diagonal batch)."""

    def __init__(self, images, device):
        if isinstance(images, dict):
            images = [images]
        self.empty = len(images) == 0
        if self.empty:
            # a data-parallel rank without an image in this step (fewer images than ranks): one image without detections --
            # zero gradients into the all-reduce, no loss terms
            images = [{"dets": np.zeros((0, 4), np.float32), "det_scores": np.zeros(0, np.float32), "det_classes": np.zeros(0, np.int32),
                       "gt_boxes": np.zeros((0, 4), np.float32), "gt_crowd": np.zeros(0, np.uint8), "gt_classes": np.zeros(0, np.int32)}]
        self.n_img = len(images)
        f32, i32 = np.float32, np.int32

        def cat(key, dtype, shape_tail):
            parts = [np.asarray(im[key], dtype=dtype).reshape((-1,) + shape_tail) for im in images if key in im]
            if len(parts) != len(images):
                return None
            return np.concatenate(parts, 0) if parts else np.zeros((0,) + shape_tail, dtype)

        dets = cat("dets", f32, (4,))
        scores = cat("det_scores", f32, ())
        classes = cat("det_classes", i32, ())
        if dets is None or scores is None or classes is None:
            raise _lib.InvalidArgumentError("batch needs dets, det_scores, det_classes")
        n_per = [np.asarray(im["dets"]).reshape(-1, 4).shape[0] for im in images]
        if scores.shape[0] != dets.shape[0] or classes.shape[0] != dets.shape[0]:
            raise _lib.InvalidArgumentError("dets / det_scores / det_classes disagree on the number of detections")
        self.det_off_h = np.concatenate([[0], np.cumsum(n_per)]).astype(i32)
        gtb = cat("gt_boxes", f32, (4,))
        self.has_gt = gtb is not None
        if self.has_gt:
            crowd = cat("gt_crowd", np.uint8, ())
            gcls = cat("gt_classes", i32, ())
            m_per = [np.asarray(im["gt_boxes"]).reshape(-1, 4).shape[0] for im in images]
            if crowd is None or gcls is None or crowd.shape[0] != gtb.shape[0] or gcls.shape[0] != gtb.shape[0]:
                raise _lib.InvalidArgumentError("gt_boxes / gt_crowd / gt_classes disagree")
        else:
            gtb = np.zeros((0, 4), f32); crowd = np.zeros(0, np.uint8); gcls = np.zeros(0, i32)
            m_per = [0] * self.n_img
        if sum(m_per) > 24576:
            raise _lib.GnetError("more than 24576 ground-truth boxes in one step: the matching kernel keeps its per-box "
                                 "state in LDS, sized by the step's total (gnet_loss returns GNET_ERR_UNSUPPORTED)")
        self.gt_off_h = np.concatenate([[0], np.cumsum(m_per)]).astype(i32)
        self.anno_off_h = np.concatenate([[0], np.cumsum(np.asarray(n_per, np.int64) * np.asarray(m_per, np.int64))]).astype(np.int64)
        self.n_det, self.n_gt = int(dets.shape[0]), int(gtb.shape[0])
        self.n_anno = int(self.anno_off_h[-1])
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        # image-feature variant: one NHWC feature map [1,H,W,C] per image (the trunk's output; caller-supplied)
        self.imfeats = None
        if all("imfeats" in im for im in images):
            self.imfeats = [torch.as_tensor(np.asarray(im["imfeats"], np.float32) if not torch.is_tensor(im["imfeats"]) else im["imfeats"])
                            .to(device=device, dtype=torch.float32).reshape((1,) + tuple(im["imfeats"].shape[-3:])).contiguous()
                            for im in images]
        self.dets, self.det_scores, self.det_classes = t(dets), t(scores), t(classes)
        self.gt_boxes, self.gt_crowd, self.gt_classes = t(gtb), t(crowd), t(gcls)
        self.det_off, self.gt_off, self.anno_off = t(self.det_off_h), t(self.gt_off_h), t(self.anno_off_h)
        self.ready = None
        if torch.device(device).type == "cuda":
            self.ready = torch.cuda.Event()                 # the uploads above are complete once this event is
            self.ready.record(torch.cuda.current_stream(device))

    def c_inputs(self):
        a = _lib.gnet_inputs()
        a.dets, a.det_scores, a.det_classes, a.det_off = _vp(self.dets), _vp(self.det_scores), _vp(self.det_classes), _vp(self.det_off)
        a.gt_boxes, a.gt_crowd, a.gt_classes = _vp(self.gt_boxes), _vp(self.gt_crowd), _vp(self.gt_classes)
        a.gt_off, a.anno_off = _vp(self.gt_off), _vp(self.anno_off)
        return a


class Gnet(object):
    name = 'gnet'
    dets = None
    det_scores = None
    det_classes = None
    gt_boxes = None
    gt_crowd = None
    gt_classes = None
    image = None

    # variable scope 'gnet' -> shared parameter storage (tf.variable_scope(reuse=True)); weak references: the storage
    # lives as long as a Gnet that uses it, not for the life of the process
    _scopes = weakref.WeakValueDictionary()

    @staticmethod
    def get_batch_spec(num_classes, is_training=True):
        """network.py:131-146."""
        spec = {
            'dets': (torch.float32, [None, 4]),
            'det_scores': (torch.float32, [None]),
            'det_classes': (torch.int32, [None]),
        }
        if is_training:
            spec.update({
                'gt_boxes': (torch.float32, [None, 4]),
                'gt_crowd': (torch.bool, [None]),
                'gt_classes': (torch.int32, [None]),
            })
        if cfg.gnet.imfeats or cfg.gnet.load_imfeats:
            spec['image'] = (torch.float32, [None, None, None, 3])
        return spec

    def __init__(self, num_classes, class_weights=None, batch=None, weight_reg=None, reuse=False,
                 device=None, imfeat_channels=1024, imfeat_stride=16):
        """cfg.gnet.imfeats=True: the batch carries `imfeats` = the trunk's NHWC feature map [1,H,W,imfeat_channels] at
        `imfeat_stride` (the reference's ResNet-101 block3 output, stride 16, network.py:52-75 -- the trunk itself is
        out of scope); crop_windows -> flatten -> reduce_imfeats FCs give block_feats[0]
        (network.py:223-240) instead of zeros, with backward into both FCs."""
        self.num_classes = num_classes
        self.multiclass = num_classes > 1
        self._imfeats = bool(cfg.gnet.imfeats)
        self.imfeat_channels, self.imfeat_stride = int(imfeat_channels), int(imfeat_stride)
        self.imfeats_need_grad = False      # also compute d loss / d imfeats (through roi_pool_grad) in run()
        if cfg.gnet.weight_init not in ('xavier', 'caffe', 'msra'):
            raise ValueError('unknown weight init {}'.format(cfg.gnet.weight_init))        # network.py:203-214
        self._lib = _lib.load()
        self.device = torch.device(device if device is not None else "cuda:0")
        g = cfg.gnet
        self._cfg = _lib.gnet_config(
            num_classes, g.num_blocks, g.neighbor_thresh, int(bool(cfg.train.normalize_loss)),
            float(cfg.train.loss_multiplyer), g.shortcut_dim, g.reduced_dim, g.pairfeat_dim, g.pwfeat_dim,
            g.pwfeat_narrow_dim, g.num_pwfeat_fc, g.predict_fc_dim, g.num_predict_fc, g.num_block_pw_fc,
            g.num_block_fc, float(g.pw_feat_multiplyer), int(bool(g.neighbor_feats)))
        n = self._lib.gnet_param_count(C.byref(self._cfg))
        if n < 0:
            _lib.check(int(n), "gnet_param_count")
        self.num_blocks = g.num_blocks
        self._spec = param_spec(num_classes, g.num_blocks)
        assert sum(int(np.prod(s)) for _, s in self._spec) == n
        self._core_n = int(n)
        # flat-buffer offset of every variable: the core tensors are packed in the C ABI's order; the reduce_imfeats
        # tensors behind them each start on a 16-byte boundary (fc.hip reads its operands as float4; the core count is
        # 1 mod 4 because the logits bias has one element).  Padding elements stay zero in params and grads.
        self._offsets, off = {}, 0
        for nm, shape in self._spec:
            self._offsets[nm] = off
            off += int(np.prod(shape))
        if self._imfeats:
            extra = imfeat_param_spec(self.imfeat_channels)
            for nm, shape in extra:
                off = (off + 3) & ~3
                self._offsets[nm] = off
                off += int(np.prod(shape))
            self._spec = self._spec + extra
            n = (off + 3) & ~3
            from .fc import FcWorkspace
            self._fc_ws = FcWorkspace(self.device)
        key = (self.name, num_classes, g.num_blocks, str(self.device), self._imfeats, self.imfeat_channels, bool(g.neighbor_feats), int(g.num_pwfeat_fc))
        if reuse:
            if key not in Gnet._scopes:
                raise ValueError("Variable scope gnet does not exist, cannot reuse")
            self.params = Gnet._scopes[key]
        else:
            self.params = self._init_params(n)
            Gnet._scopes[key] = self.params
        self.grads = torch.zeros_like(self.params)
        self.variables = {}
        self.gradients = {}
        reg = torch.zeros(n, dtype=torch.float32)
        for nm, shape in self._spec:
            k, off = int(np.prod(shape)), self._offsets[nm]
            self.variables[nm] = self.params[off:off + k].view(*shape)
            self.gradients[nm] = self.grads[off:off + k].view(*shape)
            # l2 regulariser sites: every `weights_regularizer=weight_reg` (network.py:332-403), not predict/*
            if nm.endswith("weights") and "/predict/" not in nm:
                reg[off:off + k] = 1.0
        self._reg_mask = reg.to(self.device)
        # network.py:317-322: the gnet / resnet variables minus the frozen trunk layers' prefixes (get_resnet, network.py:52-75).  The trunk
        # is the caller's here (no resnet variables live in this object), so the list starts empty; a caller that keeps trunk variables of
        # its own under the reference's names adds the first cfg.gnet.freeze_n_imfeat_layers layer prefixes and calls _filter_trainable().
        self._ignore_prefixes = []
        self._filter_trainable()
        self.weight_reg = weight_reg                       # l2 scale (tf l2_regularizer(scale): scale*sum(w^2)/2)
        if class_weights is None:
            class_weights = np.ones((num_classes + 1), dtype=np.float32)           # network.py:282-284
        self.class_weights = torch.as_tensor(np.asarray(class_weights, dtype=np.float32)).to(self.device)
        if self.class_weights.numel() != num_classes + 1:
            raise _lib.InvalidArgumentError("class_weights must have num_classes + 1 entries")
        self._ws = None
        self._buf = None
        self._shape = None
        self._row_ptr_tmp = [None, None]
        self._scratch_tmp = [None, None]
        self._side = None
        self.grad_scale = 1.0      # scale of the data-loss gradient (1 / images of the global step: mean over images)
        self.reg_scale = 1.0       # scale of the l2-regulariser gradient (1 / world size under a SUM all-reduce)
        # tests / debugging: keep the per-block pw_fc1 activations [E,64] in HBM (Gnet.debug_view("blk_h1", ...));
        # the default training path recomputes them for the rows the backward pass needs
        self.keep_edge_activations = False
        self._profiler = None
        self.transpose_after_forward = False     # (measurement switch: see _prepare_matching)
        # the 280 MB of zeroing for the backward pass (d_pw, winner maps): beside the forward pass (False) or behind it, in front of
        # the winner lists on the side stream (True).  GNET_ZERO_AFTER_FWD overrides (A/B runs).
        self.zero_after_forward = os.environ.get("GNET_ZERO_AFTER_FWD", "0") == "1"
        self._batch = batch
        if batch is not None:
            self.feed(batch)

    # ------------------------------------------------------------------ parameters
    def _init_params(self, n):
        """network.py:203-215: 'xavier' = xavier-uniform (seed cfg.random_seed), 'caffe' = variance_scaling(1.0, FAN_IN,
        uniform), 'msra' = variance_scaling(2.0, FAN_IN, truncated normal; tf.contrib.layers: stddev sqrt(1.3 f / n));
        biases constant.  (TF's RNG streams cannot be reproduced: the seed fixes OUR stream.)"""
        gen = torch.Generator().manual_seed(int(cfg.random_seed))
        flat = torch.zeros(n, dtype=torch.float32)
        for nm, shape in self._spec:
            k, off = int(np.prod(shape)), self._offsets[nm]
            if nm.endswith("weights"):
                kind = cfg.gnet.weight_init
                if kind == 'msra':
                    std = math.sqrt(1.3 * 2.0 / shape[0])
                    w = torch.empty(shape, dtype=torch.float64)
                    torch.nn.init.trunc_normal_(w, mean=0.0, std=std, a=-2 * std, b=2 * std, generator=gen)
                    flat[off:off + k] = w.to(torch.float32).reshape(-1)
                else:
                    lim = math.sqrt(6.0 / (shape[0] + shape[1])) if kind == 'xavier' else math.sqrt(3.0 * 1.0 / shape[0])
                    flat[off:off + k] = ((torch.rand(shape, generator=gen, dtype=torch.float64) * 2 - 1) * lim).to(torch.float32).reshape(-1)
            else:
                flat[off:off + k] = float(cfg.gnet.bias_const_init)
        return flat.to(self.device)

    def tensor_offsets(self):
        """Start of every variable in the flat buffers, in _spec order, plus the buffer length (the tensor table of
        gnet_clip_by_norm: tensor i = [off[i], off[i+1]) -- alignment padding belongs to the tensor in front of it and
        is zero)."""
        return [self._offsets[nm] for nm, _ in self._spec] + [int(self.params.numel())]

    def flat_to_named(self, flat):
        """Views of a flat buffer laid out like params / grads (optimizer slots), by TF variable name."""
        return {nm: flat[self._offsets[nm]:self._offsets[nm] + int(np.prod(shape))].view(*shape) for nm, shape in self._spec}

    def load_params(self, named):
        """Assign variables by TF name (e.g. from a converted checkpoint)."""
        for nm, _ in self._spec:
            if nm in named:
                self.variables[nm].copy_(torch.as_tensor(np.asarray(named[nm], dtype=np.float32)).to(self.device))

    def state_dict(self):
        return {nm: self.variables[nm].detach().cpu().numpy().copy() for nm, _ in self._spec}

    # ------------------------------------------------------------------ feeding / running
    def feed(self, batch):
        """Plays the role of feed_dict / the preloaded batch (network.py:154-160)."""
        db = batch if isinstance(batch, DeviceBatch) else DeviceBatch(batch, self.device)
        self._dbatch = db
        self.dets, self.det_scores, self.det_classes = db.dets, db.det_scores, db.det_classes
        self.gt_boxes, self.gt_crowd, self.gt_classes = db.gt_boxes, db.gt_crowd.bool(), db.gt_classes
        self.num_dets = db.n_det
        return db

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _view(self, ptr, count, dtype):
        if not ptr:
            return None
        esz = torch.empty(0, dtype=dtype).element_size()
        off = ptr - self._ws.data_ptr()
        return self._ws[off:off + count * esz].view(dtype)

    def _scoped(self, kclass, stream_ptr, fn, *args):
        """Run a C entry point that takes no gnet_buffers (the graph build) inside a profiler scope of class `kclass`."""
        prof = self._profiler
        if not prof:
            return fn(*args)
        idx = self._lib.gnet_profiler_begin(prof, _lib.KCLASSES.index(kclass), stream_ptr)
        st = fn(*args)
        self._lib.gnet_profiler_end(prof, idx, stream_ptr)
        return st

    def _count_graph(self, db):
        """Pass 1 of the graph build (per-row neighbour counts + scan) on a SIDE stream, with the edge count copied to
        pinned host memory there.  The host then waits for that small kernel only -- never for the main stream -- so
        the launches of step i+1 are issued while step i still runs (the edge count is the one value a step needs on
        the host: it sizes the workspace and the grids).  Two alternating count buffers: the main stream may still be
        reading the previous step's row_ptr copy."""
        lib = self._lib
        N = db.n_det
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
            self._e_host = [torch.zeros(1, dtype=torch.int32).pin_memory() for _ in range(2)]
            self._rp_free = [None, None]
        k = self._count_slot = (getattr(self, "_count_slot", 1) + 1) & 1
        if self._row_ptr_tmp[k] is None or self._row_ptr_tmp[k].numel() < N + 1:
            self._row_ptr_tmp[k] = torch.empty(N + 1, dtype=torch.int32, device=self.device)
            self._scratch_tmp[k] = torch.empty(N + 1024, dtype=torch.int32, device=self.device)
        thr = float(cfg.gnet.neighbor_thresh)
        if db.ready is not None:
            self._side.wait_event(db.ready)  # the inputs' upload (recorded at DeviceBatch creation) -- NOT the main stream's queue
        if self._rp_free[k] is not None:
            self._side.wait_event(self._rp_free[k])
        with torch.cuda.stream(self._side):
            s = C.c_void_p(self._side.cuda_stream)
            _lib.check(self._scoped("graph", s, lib.gnet_graph_count, _vp(db.dets), N, _vp(db.det_off), db.n_img, thr,
                                    _vp(self._row_ptr_tmp[k]), _vp(self._scratch_tmp[k]), s), "gnet_graph_count")
            if N > 0:
                self._e_host[k].copy_(self._row_ptr_tmp[k][N:N + 1], non_blocking=True)
            self._count_done = torch.cuda.Event()
            self._count_done.record(self._side)

    def _build_graph(self, db, training):
        lib, s = self._lib, self._stream()
        N = db.n_det
        thr = float(cfg.gnet.neighbor_thresh)
        k = self._count_slot
        self._count_done.synchronize()                           # the one host wait of a step: the side stream's count only
        E = int(self._e_host[k][0]) if N > 0 else 0
        torch.cuda.current_stream(self.device).wait_event(self._count_done)
        shape = _lib.gnet_shape(db.n_img, N, db.n_gt, E, db.n_anno)
        training = self._mode(training)
        if training and E > (1 << 24) - 128:
            # (before the workspace is sized for it: gnet_backward[_prepare] returns GNET_ERR_UNSUPPORTED -- the backward pass
            #  addresses the [E, 64] fp32 arrays with 32-bit byte offsets)
            raise _lib.GnetError("%d neighbour pairs in one step: a training step supports at most %d (split the batch)" % (E, (1 << 24) - 128))
        need = lib.gnet_workspace_bytes(C.byref(self._cfg), C.byref(shape), int(training))
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(int(need * 1.05) + 4096, dtype=torch.uint8, device=self.device)
        buf = _lib.gnet_buffers()
        _lib.check(lib.gnet_plan(C.byref(self._cfg), C.byref(shape), int(training), _vp(self._ws), self._ws.numel(),
                                 C.byref(buf)), "gnet_plan")
        buf.profiler = self._profiler
        self._buf, self._shape, self._training = buf, shape, training
        self._view(buf.row_ptr, N + 1, torch.int32).copy_(self._row_ptr_tmp[k][:N + 1])
        self._rp_free[k] = torch.cuda.Event()
        self._rp_free[k].record(torch.cuda.current_stream(self.device))
        _lib.check(self._scoped("graph", s, lib.gnet_graph_fill, _vp(db.dets), N, _vp(db.det_off), db.n_img, thr, buf.row_ptr,
                                buf.edge_c, buf.edge_n, buf.edge_iou, s), "gnet_graph_fill")
        self.num_edges = E
        return shape, buf

    def _prepare_matching(self, shape, inp, buf, backward=True):
        """det_anno_iou and the matching's candidate keys depend on the inputs only: they run on the side stream
        while the forward pass runs on the main one (ordered after everything the main stream has queued so far --
        the previous step's readers of these buffers -- and before gnet_loss through `_prep_done`)."""
        main = torch.cuda.current_stream(self.device)
        if getattr(self, "_prep_done", None) is None:
            self._plan_done, self._prep_done = torch.cuda.Event(), torch.cuda.Event()
        self._plan_done.record(main)
        self._side.wait_event(self._plan_done)
        with torch.cuda.stream(self._side):
            ss = C.c_void_p(self._side.cuda_stream)
            _lib.check(self._lib.gnet_match_prepare(C.byref(self._cfg), C.byref(shape), C.byref(inp), C.byref(buf), ss),
                       "gnet_match_prepare")
            self._prep_done.record(self._side)
            if shape.n_edge > 0:
                # the reverse-edge permutation is read by the backward pass only (which waits for `_bprep_done`,
                # recorded later on this stream).  Beside the forward pass by default; `transpose_after_forward` (bench.py's
                # A/B of its placement) issues it behind the forward pass instead (_prepare_backward).
                if backward and not self.transpose_after_forward:
                    _lib.check(self._scoped("graph", ss, self._lib.gnet_graph_transpose, buf.row_ptr, buf.edge_c, buf.edge_n,
                                            shape.n_edge, buf.edge_t, ss), "gnet_graph_transpose")
                if backward and not self.zero_after_forward:     # the zeroing half of the backward preparation does not need the forward pass
                    _lib.check(self._lib.gnet_backward_prepare(C.byref(self._cfg), C.byref(shape), C.byref(inp), _vp(self.params),
                                                               C.byref(buf), 1, ss), "gnet_backward_prepare")

    def _prepare_backward(self, shape, inp, buf):
        """The SegmentMax winner maps / row lists of all blocks depend on the forward pass only: they are built on the
        side stream while the main stream runs the matching, the loss and the head's backward."""
        main = torch.cuda.current_stream(self.device)
        if getattr(self, "_bprep_done", None) is None:
            self._fwd_done, self._bprep_done, self._bpos_done = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()
        self._fwd_done.record(main)
        self._side.wait_event(self._fwd_done)
        with torch.cuda.stream(self._side):
            ss = C.c_void_p(self._side.cuda_stream)
            if self.transpose_after_forward and shape.n_edge > 0:
                _lib.check(self._scoped("graph", ss, self._lib.gnet_graph_transpose, buf.row_ptr, buf.edge_c, buf.edge_n,
                                        shape.n_edge, buf.edge_t, ss), "gnet_graph_transpose")
            if self.zero_after_forward and shape.n_edge > 0:
                _lib.check(self._lib.gnet_backward_prepare(C.byref(self._cfg), C.byref(shape), C.byref(inp), _vp(self.params),
                                                           C.byref(buf), 1, ss), "gnet_backward_prepare")
            # two pieces, an event behind each: what the edge kernels read (awaited in front of the first edge_bwd_w), then the
            # reversed pairs' list positions (awaited in front of the first gather_winners: they are built beside that edge kernel)
            _lib.check(self._lib.gnet_backward_prepare(C.byref(self._cfg), C.byref(shape), C.byref(inp), _vp(self.params),
                                                       C.byref(buf), 2, ss), "gnet_backward_prepare")
            self._bprep_done.record(self._side)
            _lib.check(self._lib.gnet_backward_prepare(C.byref(self._cfg), C.byref(shape), C.byref(inp), _vp(self.params),
                                                       C.byref(buf), 3, ss), "gnet_backward_prepare")
            self._bpos_done.record(self._side)

    def _filter_trainable(self):
        """`trainable_variables` / `trainable_names` by the reference's rule (network.py:317-322)."""
        self.trainable_names = [nm for nm, _ in self._spec
                                if (nm.startswith("gnet") or nm.startswith("resnet"))
                                and not any(nm.startswith(pref) for pref in self._ignore_prefixes)]
        self.trainable_variables = [self.variables[nm] for nm in self.trainable_names]

    def _mode(self, training):
        """`training` argument of the C ABI: 0 inference, 1 training, 2 training + per-block pw_fc1 activations kept."""
        return 0 if not training else (2 if self.keep_edge_activations else 1)

    def defer_backward_until(self, event):
        """The next backward pass (the next writer of `grads`) waits for `event` on its stream -- the graph build, the forward
        pass and the loss of that step do not.  bench.py --gpus N hands the all-reduce's completion event here."""
        self._grads_busy = event

    def begin(self, batch=None):
        """First, asynchronous half of run(): feed + neighbour counting.  Several Gnets (sharing variables
        through reuse=True, each on its own stream) can be begun before any of them is finished, so that the
        one host sync of a step (reading the edge count) of one lane overlaps the kernels of the others."""
        db = self.feed(batch) if batch is not None else self._dbatch
        with torch.cuda.device(self.device):
            self._count_graph(db)
        self._begun = True
        return self

    def run(self, batch=None, training=None, backward=None):
        """One evaluation of the graph = the reference's sess.run.  training=None -> GT present."""
        with torch.cuda.device(self.device):       # kernels launch on the current device: make it this net's
            return self._run(batch, training, backward)

    def _run(self, batch, training, backward):
        if batch is not None or not getattr(self, "_begun", False):
            self.begin(batch)
        self._begun = False
        db = self._dbatch
        if training is None:
            training = db.has_gt
        if backward is None:
            backward = training
        if training and not db.has_gt:
            raise _lib.InvalidArgumentError("training needs gt_boxes / gt_crowd / gt_classes")
        lib, s = self._lib, self._stream()
        shape, buf = self._build_graph(db, training)
        inp = db.c_inputs()
        self._inputs = inp
        if self._imfeats:
            if db.empty:
                # a rank without an image in this step: no detections, so no start features to compute (the placeholder image
                # carries no feature map) -- and the reduce_imfeats gradients below are exactly zero
                self.roifeats = self.det_imfeats = None
                self._imfeat_acts = []
                buf.start_feat = _vp(self.params)          # [0,128]: never read
            else:
                buf.start_feat = _vp(self._imfeat_forward(db))
        if training:
            self._prepare_matching(shape, inp, buf, backward)
        _lib.check(lib.gnet_forward(C.byref(self._cfg), C.byref(shape), C.byref(inp), _vp(self.params),
                                    C.byref(buf), self._mode(training), s), "gnet_forward")
        if training and backward:
            self._prepare_backward(shape, inp, buf)
        if training:
            torch.cuda.current_stream(self.device).wait_event(self._prep_done)
            _lib.check(lib.gnet_loss(C.byref(self._cfg), C.byref(shape), C.byref(inp), _vp(self.class_weights),
                                     float(self.grad_scale), C.byref(buf), 1, s), "gnet_loss")
            if backward:
                if getattr(self, "_grads_busy", None) is not None:
                    # a collective launched on another stream still reads the gradient buffer of the previous step
                    # (data_parallel.GradientExchange.launch): everything up to here ran beside it
                    torch.cuda.current_stream(self.device).wait_event(self._grads_busy)
                    self._grads_busy = None
                # (the main stream waits for the winner lists inside gnet_backward, where they are first needed)
                _lib.check(lib.gnet_backward(C.byref(self._cfg), C.byref(shape), C.byref(inp), _vp(self.params),
                                             C.byref(buf), _vp(self.grads), 1, C.c_void_p(self._bprep_done.cuda_event),
                                             C.c_void_p(self._bpos_done.cuda_event), s),
                           "gnet_backward")
                if self._imfeats:
                    if db.empty:
                        for nm, _ in self._spec:
                            if nm.startswith("gnet/reduce_imfeats/"):
                                self.gradients[nm].zero_()
                        self.imfeats_grad = [] if self.imfeats_need_grad else None
                    else:
                        self._imfeat_backward(db)
                if self.weight_reg:
                    # slim get_total_loss adds sum(l2_regularizer(scale)(w)) -> d/dw = scale * w (train.py:231-238),
                    # ONCE per optimisation step whatever the number of images in it: not scaled by grad_scale.
                    # Under data parallelism every rank adds reg_scale = 1 / world of it (the all-reduce sums).
                    self.grads.addcmul_(self.params, self._reg_mask, value=float(self.weight_reg) * float(self.reg_scale))
        return self

    # ------------------------------------------------------------------ image-feature start features
    def _imfeat_layers(self):
        names = [nm[:-len("weights")] for nm, _ in self._spec if nm.startswith("gnet/reduce_imfeats/") and nm.endswith("weights")]
        return names        # one or two scopes, in order

    def _imfeat_forward(self, db):
        """network.py:223-240: roifeats = crop_windows(imfeats, dets) [N,7,7,C]; det_imfeats = flatten -> FC(s), ReLU."""
        from .fc import fc_forward
        from .roi_pooling_layer.roi_pooling_op import roi_pool_raw
        if db.imfeats is None:
            raise _lib.InvalidArgumentError("cfg.gnet.imfeats=True: every image of the batch needs `imfeats` [1,H,W,C]")
        tops, self._roi_argmax, self._frcn_boxes = [], [], []
        for i, fm in enumerate(db.imfeats):
            if fm.shape[-1] != self.imfeat_channels:
                raise _lib.InvalidArgumentError("imfeats has %d channels, Gnet was built for %d" % (fm.shape[-1], self.imfeat_channels))
            d = db.dets[int(db.det_off_h[i]):int(db.det_off_h[i + 1])]
            boxes = to_frcn_coords(enlarge_windows(d))
            top, am = roi_pool_raw(fm, boxes, int(cfg.imfeat_crop_height), int(cfg.imfeat_crop_width), 1.0 / self.imfeat_stride)
            tops.append(top); self._roi_argmax.append(am); self._frcn_boxes.append(boxes)
        self.roifeats = torch.cat(tops, 0) if len(tops) > 1 else tops[0]
        self.det_imfeats = self.roifeats.reshape(self.roifeats.shape[0], -1)          # tf.contrib.layers.flatten
        x, acts = self.det_imfeats, []
        for scope in self._imfeat_layers():
            x = fc_forward(x, self.variables[scope + "weights"], self.variables[scope + "biases"], True, self._fc_ws)
            acts.append(x)
        self._imfeat_acts = acts
        return acts[-1]

    def _imfeat_backward(self, db):
        from .fc import fc_backward
        N = self._shape.n_det
        dy = self._view(self._buf.d_x, N * 128, torch.float32).view(N, 128)           # gradient wrt the start features
        scopes = self._imfeat_layers()
        inputs = [self.det_imfeats] + self._imfeat_acts[:-1]
        for li in range(len(scopes) - 1, -1, -1):
            need_dx = li > 0 or self.imfeats_need_grad
            dy = fc_backward(inputs[li], self.variables[scopes[li] + "weights"], self._imfeat_acts[li], dy, True,
                             self.gradients[scopes[li] + "weights"], self.gradients[scopes[li] + "biases"], self._fc_ws, need_dx)
        self.imfeats_grad = None
        if self.imfeats_need_grad:
            from .roi_pooling_layer.roi_pooling_op import roi_pool_grad
            g = dy.view(-1, int(cfg.imfeat_crop_height), int(cfg.imfeat_crop_width), self.imfeat_channels)
            self.imfeats_grad = []
            for i, fm in enumerate(db.imfeats):
                lo, hi = int(db.det_off_h[i]), int(db.det_off_h[i + 1])
                self.imfeats_grad.append(roi_pool_grad(fm, self._frcn_boxes[i], self._roi_argmax[i], g[lo:hi].contiguous(),
                                                       int(cfg.imfeat_crop_height), int(cfg.imfeat_crop_width), 1.0 / self.imfeat_stride))

    # ------------------------------------------------------------------ measurement
    @staticmethod
    def count_edges(dets, device):
        """Number of neighbour pairs (incl. self pairs) of one image: pass 1 of the graph build only.  The cost of an
        image is ~ its edge count, not its detection count: used to balance images over ranks."""
        lib = _lib.load()
        device = torch.device(device)
        d = torch.as_tensor(np.ascontiguousarray(np.asarray(dets, np.float32).reshape(-1, 4))).to(device)
        n = int(d.shape[0])
        if n == 0:
            return 0
        off = torch.tensor([0, n], dtype=torch.int32, device=device)
        row_ptr = torch.empty(n + 1, dtype=torch.int32, device=device)
        scratch = torch.empty(n + 1024, dtype=torch.int32, device=device)
        with torch.cuda.device(device):
            s = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
            _lib.check(lib.gnet_graph_count(_vp(d), n, _vp(off), 1, float(cfg.gnet.neighbor_thresh), _vp(row_ptr), _vp(scratch), s),
                       "gnet_graph_count")
        return int(row_ptr[n].item())

    def release_workspace(self):
        self._ws = None
        self._buf = None

    def backward_stats(self):
        """Sizes the sparse backward pass worked on in the last run (device counters read back; measurement only):
        mean winner rows per block, rows of the pw-MLP backward, bytes reduce_partials read."""
        E, N, B = int(self.num_edges), int(self.num_dets), self.num_blocks
        if E == 0:
            return {"winners_per_block": 0.0, "pw_rows": 0, "arena_bytes_read": 0}
        n_words = (E + 63) // 64
        n_wg = (n_words + 1 + 255) // 256
        rl = self.debug_view("rl_scratch", (B + 1) * (2 * n_wg + 1), dtype=torch.int32).cpu().numpy()
        off = rl[(B + 1) * n_wg:].reshape(B + 1, n_wg + 1)
        totals = off[:, n_wg]
        cp = self.num_classes if self.num_classes > 1 else 1
        n = int(self.params.numel())
        g_w1c = max(1, min(128, 2048 // (2 * cp))); g_w1 = max(1, min(512, (N + 3) // 4)); g_pw = min((E + 31) // 32, 256)
        tiles = (N + 31) // 32
        g_edge, g_node, g_head = 512, min(tiles, 256), min(tiles, 256)                        # the grids of gnet_backward
        w1c = 2 * cp * 256; pw1 = (2 * cp + 7) * 256 + 256; pw = pw1 + 256 * 256 + 256 + 256 * 32 + 32
        blk = 128 * 32 + 32 + 96 * 64 + 64 + 64 * 64 + 64 + 64 * 64 + 64 + 64 * 128 + 128
        if cfg.gnet.neighbor_feats:
            blk += 128 * 32 + 32
        edge_p = 32 * 64 + 64 * 64 + 64
        head = n - pw - B * blk
        reads = w1c * g_w1c + (pw1 - w1c) * g_w1 + (pw - pw1) * g_pw + B * (edge_p * g_edge + (blk - edge_p) * g_node) + head * g_head
        return {"winners_per_block": float(totals[:B].mean()), "pw_rows": int(totals[B]), "arena_bytes_read": int(reads) * 4}

    def enable_kernel_timing(self, classes=None, capacity=4096, stride=1):
        """HIP-event timing of the selected kernel classes (names in _lib.KCLASSES) on the launch stream; stride > 1
        brackets only every stride-th launch of a class (a sample: an event pair costs the stream a few microseconds)."""
        mask = 0
        for i, nm in enumerate(_lib.KCLASSES):
            if classes is None or nm in classes:
                mask |= 1 << i
        if self._profiler:
            self._lib.gnet_profiler_destroy(self._profiler)
        out = C.c_void_p()
        _lib.check(self._lib.gnet_profiler_create(capacity, mask, C.byref(out)), "gnet_profiler_create")
        self._profiler = out.value
        if stride != 1:
            _lib.check(self._lib.gnet_profiler_set_stride(self._profiler, int(stride)), "gnet_profiler_set_stride")

    def read_kernel_timing(self):
        ms = (C.c_double * len(_lib.KCLASSES))()
        cnt = (C.c_int32 * len(_lib.KCLASSES))()
        _lib.check(self._lib.gnet_profiler_read(self._profiler, ms, cnt), "gnet_profiler_read")
        return {nm: (ms[i], cnt[i]) for i, nm in enumerate(_lib.KCLASSES) if cnt[i]}

    # ------------------------------------------------------------------ outputs (Gnet attributes)
    @property
    def prediction(self):
        return self._view(self._buf.prediction, self._shape.n_det, torch.float32)

    @property
    def labels(self):
        return self._view(self._buf.labels, self._shape.n_det, torch.float32)

    @property
    def weights(self):
        return self._view(self._buf.weights, self._shape.n_det, torch.float32)

    @property
    def det_gt_matching(self):
        return self._view(self._buf.det_gt_matching, self._shape.n_det, torch.int32)

    @property
    def det_anno_iou(self):
        db = self._dbatch
        flat = self._view(self._buf.det_anno_iou, max(self._shape.n_anno, 0), torch.float32)
        outs = []
        for i in range(db.n_img):
            n = int(db.det_off_h[i + 1] - db.det_off_h[i]); m = int(db.gt_off_h[i + 1] - db.gt_off_h[i])
            outs.append(flat[int(db.anno_off_h[i]):int(db.anno_off_h[i + 1])].view(n, m))
        return outs[0] if db.n_img == 1 else outs

    @property
    def image_losses(self):
        """[n_img, 2]: (loss_unnormed, loss_normed) of every image of the batch."""
        return self._view(self._buf.loss, 2 * self._shape.n_img, torch.float32).view(-1, 2)

    @property
    def loss_unnormed(self):
        return self.image_losses[:, 0].sum()

    @property
    def loss_normed(self):
        return self.image_losses[:, 1].sum()

    @property
    def loss(self):
        l = self.loss_normed if cfg.train.normalize_loss else self.loss_unnormed
        return l * float(cfg.train.loss_multiplyer)

    def regularization_loss(self):
        if not self.weight_reg:
            return torch.zeros((), device=self.device)
        return 0.5 * float(self.weight_reg) * (self.params * self.params * self._reg_mask).sum()

    @property
    def pw_feats(self):
        """Gnet.pw_feats (network.py:221): the pairwise-feature MLP's output [E,32] -- or, with cfg.gnet.num_pwfeat_fc = 0,
        the raw _geometry_feats columns [E, 2C'+7] x pw_feat_multiplyer themselves (assembled on demand from the geometry
        columns the kernels keep and the one-hot x score columns, network.py:413-419; the hot path never materialises them)."""
        E = self._shape.n_edge
        if self._cfg.num_pwfeat_fc > 0:
            return self._view(self._buf.pw_feats, E * 32, torch.float32).view(-1, 32)
        geo = self._view(self._buf.geo, E * 8, torch.float32).view(-1, 8)[:, :7]
        db = self._dbatch
        N = db.n_det
        mult = float(self._cfg.pw_feat_multiplyer)
        if self._cfg.num_classes > 1:
            Cn = self._cfg.num_classes
            cls = db.det_classes.long() - 1
            ok = (cls >= 0) & (cls < Cn)
            sc = torch.zeros(N, Cn, dtype=torch.float32, device=self.device)
            idx = torch.nonzero(ok).view(-1)
            sc[idx, cls[idx]] = db.det_scores[idx] * mult
        else:
            sc = (db.det_scores * mult).view(-1, 1)
        pairs = self.neighbor_pair_idxs
        return torch.cat([sc[pairs[:, 0]], sc[pairs[:, 1]], geo], 1)

    @property
    def block_feats(self):
        N = self._shape.n_det
        out = [self._imfeat_acts[-1] if self._imfeats else torch.zeros(N, 128, device=self.device)]
        if not self._training:
            out += [None] * (self.num_blocks - 1)
            out.append(self._view(self._buf.block_feats[self.num_blocks], N * 128, torch.float32).view(N, 128))
            return out
        for k in range(1, self.num_blocks + 1):
            out.append(self._view(self._buf.block_feats[k], N * 128, torch.float32).view(N, 128))
        return out

    @property
    def neighbor_pair_idxs(self):
        E = self._shape.n_edge
        c = self._view(self._buf.edge_c, E, torch.int32)
        n = self._view(self._buf.edge_n, E, torch.int32)
        return torch.stack([c, n], 1).long()

    @property
    def det_det_iou(self):
        """network.py:176 -- the dense [N, N] IoU matrix of a single-image batch, materialised on demand from the boxes
        by the same kernel that builds det_anno_iou (no caller of the reference reads it; the hot path keeps only the
        values at the neighbour pairs, `edge_iou`)."""
        db = self._dbatch
        if db.n_img != 1:
            raise _lib.GnetError("det_det_iou is defined per image: feed one image")
        n = db.n_det
        out = torch.empty(n, n, dtype=torch.float32, device=self.device)
        if n:
            off = torch.tensor([0, n], dtype=torch.int32, device=self.device)
            aoff = torch.tensor([0, n * n], dtype=torch.int64, device=self.device)
            with torch.cuda.device(self.device):
                _lib.check(self._lib.gnet_box_iou(_vp(db.dets), n, _vp(db.dets), n, _vp(off), _vp(off), _vp(aoff), 1, _vp(out),
                                                  self._stream()), "gnet_box_iou")
        return out

    @property
    def edge_iou(self):
        return self._view(self._buf.edge_iou, self._shape.n_edge, torch.float32)

    @property
    def row_ptr(self):
        return self._view(self._buf.row_ptr, self._shape.n_det + 1, torch.int32)

    def debug_view(self, name, count, dtype=torch.float32, index=None):
        p = getattr(self._buf, name)
        if index is not None:
            p = p[index]
        return self._view(p, count, dtype)


# ---------------------------------------------------------------------------------------------
# Thin helpers of the image-feature variant (network.py:78-118).  The ResNet trunk itself is out of scope
# (SURVEY §2 row 12); these produce the RoI-pooled detection features from a caller-supplied feature map.
def enlarge_windows(dets, padding=0.5):
    """network.py:78-86: pad every box by `padding` of its width/height on each side (xyxy in, xyxy out)."""
    x1, y1, x2, y2 = dets[:, 0:1], dets[:, 1:2], dets[:, 2:3], dets[:, 3:4]
    w, h = x2 - x1, y2 - y1
    cx, cy = (x1 + x2) / 2.0, (y1 + y2) / 2.0
    nw2, nh2 = w * (0.5 + padding), h * (0.5 + padding)
    return torch.cat([cx - nw2, cy - nh2, cx + nw2, cy + nh2], 1)


def to_frcn_coords(boxes):
    """network.py:97-100: prepend the batch index 0."""
    return torch.cat([torch.zeros(boxes.shape[0], 1, dtype=boxes.dtype, device=boxes.device), boxes], 1)


def crop_windows(imfeats, dets, stride):
    """network.py:103-118: RoI max pooling of the enlarged detection windows, NHWC feature map.
    Returns (detection_feats [N, crop_h, crop_w, C], frcn_boxes [N,5]); differentiable w.r.t. imfeats."""
    from .roi_pooling_layer.roi_pooling_op import roi_pool
    frcn_boxes = to_frcn_coords(enlarge_windows(dets))
    feats, _ = roi_pool(imfeats, frcn_boxes, pooled_height=cfg.imfeat_crop_height,
                        pooled_width=cfg.imfeat_crop_width, spatial_scale=1.0 / stride)
    return feats, frcn_boxes
