"""Multi-view fit of ONE shared mesh, sharded by camera across the GPUs of a node.

Every rank holds the whole mesh (vertices, triangles, textures: a few MB) and a contiguous slice of
the cameras; it renders its views, evaluates its share of the loss and back-propagates to per-vertex /
per-texel gradients already summed over its views.  The only exchange is ONE all-reduce(SUM) of a flat
f32 buffer [3V (+ F*ts^3*3)] per step -- RCCL over xGMI when the process group's backend is "nccl".
The reference has no such path (it only runs a different image per rank); this is the north-star's
"shard by camera + all-reduce of vertex gradients".
"""
import os
import weakref

import torch
import torch.distributed as dist

from . import _lib, neural_renderer as nr
from .graph import CapturedStep
from .core.losses import multiview_fit_loss, photometric_loss, silhouette_loss


def shard_views(n_views, rank, world_size):
    """Contiguous split of the cameras: rank r owns views [lo, hi).  A camera count that does not divide leaves its
    remainder with the LOW ranks (the first n % R ranks take one camera more), so shard sizes differ by at most one and
    rank 0 always holds a largest shard.  Every rank needs at least one camera."""
    if world_size <= 0 or not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} of {world_size}")
    if n_views < world_size:
        raise ValueError(f"{n_views} views cannot be shared among {world_size} ranks (every rank renders at least one)")
    per, extra = divmod(n_views, world_size)
    lo = rank * per + min(rank, extra)
    return lo, lo + per + (1 if rank < extra else 0)


COLLECTIVES_WITH_ONE_RANK = False      # debug (bench.py D3M_BENCH_FORCE_DIST): issue the collective even in a group of one


def allreduce_sum_(flat, group=None):
    """In-place SUM-all-reduce of one flat device buffer (one collective per step).  Under "nccl" (= RCCL) the buffer
    stays on the device and travels over xGMI; under "gloo" (CPU-test / single-GPU debug configuration only) it is
    staged through the host."""
    if not (dist.is_available() and dist.is_initialized()):
        return flat
    if dist.get_world_size(group) == 1 and not COLLECTIVES_WITH_ONE_RANK:
        return flat
    if flat.is_cuda and dist.get_backend(group) == "gloo":
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        flat.copy_(host)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def allreduce_sum_start(flat, group=None):
    """Start the in-place SUM-all-reduce of a flat device buffer and return a handle for allreduce_wait() -- None when
    there is nothing to wait for (no process group, one rank, or the gloo debug configuration, which is staged through
    the host synchronously).  Under "nccl" (= RCCL) the collective runs on the communicator's own stream, ordered behind
    what the CURRENT stream has been given so far: kernels issued afterwards run beside it."""
    if not (dist.is_available() and dist.is_initialized()):
        return None
    if dist.get_world_size(group) == 1 and not COLLECTIVES_WITH_ONE_RANK:
        return None
    if flat.is_cuda and dist.get_backend(group) == "gloo":
        allreduce_sum_(flat, group)
        return None
    return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)


def allreduce_wait(work):
    """The current stream waits for a collective started by allreduce_sum_start (no host synchronisation under RCCL)."""
    if work is not None:
        work.wait()


def allreduce_flat(tensors, group=None):
    """SUM-all-reduce several tensors as one flat buffer (allocates the buffer: for one-off exchanges; the step itself
    uses the persistent buffer of MultiViewFit)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return tensors
    flat = allreduce_sum_(torch.cat([t.reshape(-1) for t in tensors]), group)
    out, o = [], 0
    for t in tensors:
        out.append(flat[o:o + t.numel()].view_as(t))
        o += t.numel()
    return out


class MultiViewFit:
    """One optimisation step = render local views -> loss vs targets -> backward -> all-reduce.

    vertices [V,3], triangles [F,3] int32, textures [F,ts,ts,ts,3], eyes [n_views,3] (look_at cameras).
    Loss: photometric_loss(rgb, rgb_t, mask=alpha_t) + sum((alpha-alpha_t)^2)/P
                         + photometric_loss(depth, depth_t, mask=alpha_t),  P = pixels per view,
    taken over ALL n_views cameras.  Sharded over R ranks the objective stays that one: the photometric terms are
    means over the mask of all cameras, so every rank normalises its sums by the GLOBAL sum(mask) -- a constant of the
    targets, all-reduced once in set_targets_from() -- and the ranks' values and gradients then simply add up
    (R ranks x n/R cameras == 1 rank x n cameras, tests/test_gpu_multirank.py).  Per step ONE all-reduce(SUM) of the
    persistent flat buffer [loss | grad_vertices 3V | grad_textures 3 F ts^3].
    """

    def __init__(self, vertices, triangles, textures, eyes, image_size=512, anti_aliasing=False, rank=0,
                 world_size=1, optimise_textures=True, device="cuda", objective_in_renderer=True, view_groups=1,
                 split_exchange=None, loss_form="linked"):
        self.device = torch.device(device)
        self.rank, self.world_size = rank, world_size
        lo, hi = shard_views(len(eyes), rank, world_size)
        self.n_local = hi - lo
        self.vertices = torch.as_tensor(vertices, dtype=torch.float32).to(self.device).requires_grad_(True)
        self.triangles = torch.as_tensor(triangles, dtype=torch.int32).to(self.device)
        self.textures = torch.as_tensor(textures, dtype=torch.float32).to(self.device).requires_grad_(optimise_textures)
        self.eyes = torch.as_tensor(eyes, dtype=torch.float32)[lo:hi].to(self.device).contiguous()
        self.renderer = nr.Renderer(image_size=image_size, anti_aliasing=anti_aliasing, camera_mode="look_at",
                                    fill_back=True)
        self.renderer.eye = self.eyes
        self.renderer.view_groups = view_groups     # concurrent pipelines over this rank's views (rasterize.py)
        self.image_size = image_size
        self.objective_in_renderer = objective_in_renderer      # False: render() the images, then loss() on them
        # ... and HOW, when the objective is not evaluated inside the renderer:
        #   "linked"     Renderer.fit_targets registered + multiview_fit_loss on the images: value and walk records come out
        #                of the render pass (the drop-in form with two added lines: DESIGN.md 4.4);
        #   "operators"  what a caller of the REFERENCE's surface runs unmodified: Renderer.render() with nothing
        #                registered, the objective composed from the package's drop-ins for deep3dmap's own loss functions
        #                (photometric_loss, utils.py:105-114; silhouette_loss, examples/example2.py:46) -- three loss
        #                nodes, gradient IMAGES back through the output epilogue's adjoint;
        #   "torch"      the same composition in plain torch operators (as models/frameworks/gan2shape.py:469-497 and
        #                examples/example2.py:43-47 write their losses).
        if loss_form not in ("linked", "operators", "torch"):
            raise ValueError("loss_form: 'linked', 'operators' or 'torch'")
        self.loss_form = loss_form
        self.keep_images = False        # True: the fused objective's pass also writes this step's images to self.images
        self.images = None              # (rgb, depth, alpha)
        self.targets = None
        self.mask_sum = None            # [1] device scalar: sum of the mask over ALL ranks' views
        self._mask_sum_local = None
        # the step's results, packed where they are produced (inside the captured step): [loss | grad_v | grad_t]
        n_t = self.textures.numel() if optimise_textures else 0
        self._flat = torch.zeros(1 + self.vertices.numel() + n_t, dtype=torch.float32, device=self.device)
        # ... by the rendering node itself where it can: the node's backward writes the two gradients, its forward the loss,
        # straight into these views of the flat buffer (render_fit_loss(grad_sink=...)); _forward_backward packs only what did not
        nv = self.vertices.numel()
        self._sink = (self._flat[1:1 + nv].view(1, *self.vertices.shape),
                      self._flat[1 + nv:].view(1, *self.textures.shape) if n_t else None, self._flat[0:1])
        self._use_sink = False          # only the step's own forward + backward hands the buffers to the node (per call)
        # eager or replayed, always on one stream.  (Through a weak reference: a bound method would make fit -> runner ->
        # fit a cycle, and a fit with its captured graph, memory pool and streams should die with its last reference, not
        # whenever the collector next runs -- see graph.py on collections inside a capture.)
        me = weakref.ref(self)
        # SPLIT EXCHANGE (SURVEY 8e "visible at small per-GPU work"): of the 10.2 MB a headline step exchanges, 9.6 MB are
        # the texture gradient, and that is final as soon as the gathered texture / depth pass and the sum over views have
        # run -- before the edge gradient (the longest kernel chain of the step) starts.  The step is then TWO parts (two
        # HIP graphs when captured): forward + texture side of backward | all-reduce of the texture part STARTED on the
        # communicator's stream | geometry side of backward beside it | all-reduce of [loss | vertex gradient].  The node's
        # two halves are called by hand (rasterize.LitFitManual: a capture cannot end on autograd's worker thread).
        # Default: on when gradients are exchanged, the fused objective runs (one pipeline, look_at cameras) AND the
        # rank's batch is a big one (d3m_forward_big_batch: every kernel fills the chip by itself).  Measured on one rank
        # through RCCL (profiles/r06_bench_other_configs.jsonl): the split form costs a 32-view step 0.03 ms of graph launches
        # and collectives plus the 0.05 ms its side branches buy since round 6 (the two halves run on one stream each), and
        # hides a 9.6 MB all-reduce (0.07-0.17 ms over xGMI); an 8-view shard it costs 0.08 ms, which is about what its
        # collective takes -- no gain there, so the small shards keep the one-graph step.
        if split_exchange is None:
            # (decided on the LARGEST shard, rank 0's: the ranks of one job must agree on the form of the exchange --
            #  with shards of unequal size a per-rank decision could pair one rank's two collectives with another's one)
            n_largest = shard_views(len(eyes), 0, world_size)[1]
            env = os.environ.get("D3M_SPLIT_EXCHANGE", "1")          # "0": never, "force": whatever the batch (tests)
            split_exchange = (env != "0" and (world_size > 1 or COLLECTIVES_WITH_ONE_RANK) and
                              (env == "force" or
                               _lib.lib().d3m_forward_big_batch(int(n_largest), int(self.triangles.shape[0]),
                                                                int(image_size * (2 if anti_aliasing else 1))) == 1))
        self.split_exchange = bool(split_exchange and objective_in_renderer and optimise_textures and view_groups == 1
                                   and self.renderer._on_the_fly())
        self._manual = self._tex_work = None
        self._grad_one = torch.ones(1, dtype=torch.float32, device=self.device)
        if self.split_exchange:
            self._one = torch.ones(1, dtype=torch.float32, device=self.device)
            self._runner = CapturedStep([lambda: me()._forward_and_texture_side(), lambda: me()._geometry_side()],
                                        between=[lambda: me()._start_texture_exchange()])
        else:
            self._runner = CapturedStep(lambda: me()._forward_backward())

    def render(self, vertices=None, textures=None):
        v = self.vertices if vertices is None else vertices
        t = self.textures if textures is None else textures
        # the drop-in form (render, then the objective on the images): the objective is REGISTERED with the renderer, whose
        # pass then leaves value and walk records behind for multiview_fit_loss to find (Renderer.fit_targets)
        self.renderer.fit_targets = None
        if not self.objective_in_renderer and self.loss_form == "linked" and self.targets is not None and vertices is None:
            rgb_t, depth_t, alpha_t = self.targets
            self.renderer.fit_targets = (rgb_t, depth_t, alpha_t, alpha_t, self.mask_sum)
        # one mesh, one texture set, n_local cameras (renderer.eye is [n_local, 3]): batch-1 inputs are shared
        return self.renderer(v[None], self.triangles[None], t[None])

    @torch.no_grad()
    def set_targets_from(self, target_vertices):
        tv = torch.as_tensor(target_vertices, dtype=torch.float32).to(self.device)
        rgb, depth, alpha = self.render(vertices=tv)
        self.targets = (rgb.detach(), depth.detach(), alpha.detach())
        # the objective's mask is alpha_t: its sum over every rank's cameras is a constant of the targets (with it known
        # before the render, the fused objective leaves its gradient ready for the edge gradient: rasterize.py)
        self._mask_sum_local = self.targets[2].sum().reshape(1)
        self.mask_sum = allreduce_sum_(self._mask_sum_local.clone())

    def loss(self, rgb, depth, alpha, fused=True):
        """The fit objective.  `fused=False` composes it from the three loss operators (the definition; the fused
        node computes the same value and gradients in 3 launches instead of ~11)."""
        rgb_t, depth_t, alpha_t = self.targets
        if fused and self.loss_form == "linked":
            return multiview_fit_loss(rgb, depth, alpha, rgb_t, depth_t, alpha_t, alpha_t, self.mask_sum)
        mask = alpha_t[:, None]
        pixels = float(self.image_size * self.image_size)
        # photometric_loss divides by the LOCAL sum(mask); as this rank's part of the global mean it weighs local/global
        share = 1.0 if self.mask_sum is None else (self._mask_sum_local / self.mask_sum)[0]
        if self.loss_form == "torch":       # deep3dmap's own formulas, eager torch (utils.py:105-114; example2.py:46)
            den = mask.sum()
            l_rgb = ((rgb - rgb_t).abs() * mask.expand_as(rgb)).sum() / (3.0 * den)
            l_depth = ((depth - depth_t).abs()[:, None] * mask).sum() / den
            return l_rgb * share + ((alpha - alpha_t) ** 2).sum() / pixels + l_depth * share
        return (photometric_loss(rgb, rgb_t, mask=mask) * share + silhouette_loss(alpha, alpha_t) / pixels +
                photometric_loss(depth[:, None], depth_t[:, None], mask=mask) * share)

    def fit_loss(self):
        """The objective of the current mesh against the targets, evaluated inside the rendering node when the renderer
        allows it (no images, no image gradients in memory); otherwise render() + loss()."""
        r = self.renderer
        if self.objective_in_renderer and r._on_the_fly():
            rgb_t, depth_t, alpha_t = self.targets
            if self.keep_images and self.images is None:        # persistent buffers (a captured step writes in place)
                self.images = tuple(torch.empty_like(t) for t in (rgb_t, depth_t, alpha_t))
            return r.render_fit_loss(self.vertices[None], self.triangles[None], self.textures[None],
                                     (rgb_t, depth_t, alpha_t, alpha_t, self.mask_sum),
                                     images_out=self.images if self.keep_images else None,
                                     grad_sink=self._sink if self._use_sink else None)
        return self.loss(*self.render())

    def _forward_backward(self):
        self.vertices.grad = None
        self.textures.grad = None
        # backward runs right behind forward, on the same stream and inside the same capture: only HERE may the forward
        # leave its side branch (visibility list, edge plan, the loss's last reduction step) open for backward to join.
        # A bare fit_loss() (logging, evaluation) joins everything before it returns.
        # Only here, too, does the node get the flat buffer's views as its destinations (a bare fit_loss() returns a loss
        # of its own, not an alias of _flat[0] that the next step's collective overwrites).
        self.renderer.defer_plan_join = True
        self._use_sink = True
        try:
            loss = self.fit_loss()
            loss.backward(self._grad_one.reshape(loss.shape))   # (a constant: autograd's implicit ones_like is a fill launch)
        finally:
            self.renderer.defer_plan_join = False
            self._use_sink = False
        # [loss | grad_v | grad_t] in the persistent buffer the collective runs on: already there when the rendering node
        # produced them in place (grad_sink); packed here (part of the captured step) otherwise
        parts = [loss.detach().reshape(1), self.vertices.grad.reshape(-1)]
        if self.textures.requires_grad:
            parts.append(self.textures.grad.reshape(-1))
        at = 0
        for part in parts:
            dst = self._flat[at:at + part.numel()]
            if part.data_ptr() != dst.data_ptr():
                dst.copy_(part)
            at += part.numel()
        return self._flat

    # ---- the step in two parts (split_exchange) --------------------------------------------------------------------
    def _forward_and_texture_side(self):
        from .neural_renderer.rasterize import LitFitManual
        rgb_t, depth_t, alpha_t = self.targets
        if self.keep_images and self.images is None:
            self.images = tuple(torch.empty_like(t) for t in (rgb_t, depth_t, alpha_t))
        self._manual = LitFitManual(vertices_grad=True, textures_grad=True)
        # (the loss lands in _flat[0], the gradients in their places behind it: the node's grad_sink)
        self.renderer.render_fit_loss_manual(self._manual, self.vertices[None], self.triangles[None], self.textures[None],
                                             (rgb_t, depth_t, alpha_t, alpha_t, self.mask_sum),
                                             images_out=self.images if self.keep_images else None, grad_sink=self._sink)
        self._manual.backward_texture_side(self._one)

    def _start_texture_exchange(self):
        """Between the step's two parts, never inside a capture.  One texture all-reduce is in flight at a time: a
        collective still pending from the previous call (warm-up iterations, the call between the two captures) is
        waited for before the buffer is handed to the next one."""
        nv = self.vertices.numel()
        allreduce_wait(self._tex_work)
        self._tex_work = allreduce_sum_start(self._flat[1 + nv:])

    def _geometry_side(self):
        gv, gt = self._manual.backward_geometry_side()
        self._manual = None
        nv = self.vertices.numel()
        for part, dst in ((gv, self._flat[1:1 + nv]), (gt, self._flat[1 + nv:])):
            if part.data_ptr() != dst.data_ptr():       # (not reached with a valid grad_sink; kept for safety)
                dst.copy_(part.reshape(-1))
        return self._flat

    def capture_graph(self, warmup=3):
        """Capture forward + loss + backward of one step into a HIP graph (see deep3dmap_amd/graph.py).
        Vertices / textures / targets are updated IN PLACE between replays."""
        self.vertices.grad = None
        self.textures.grad = None
        self._runner.capture(warmup)
        if self.split_exchange:
            # capture() ran the callback between the parts for real (warm-up, and once between the two captures): that
            # last in-place all-reduce of the texture part must have finished on EVERY rank before the first replay of
            # part A writes the buffer again -- otherwise, under rank skew, it sums (and overwrites) the fresh gradient
            allreduce_wait(self._tex_work)
            self._tex_work = None
            torch.cuda.synchronize(self.device)
        return self

    def release_graph(self):
        self._runner.release()

    @property
    def stream(self):
        """The stream every step of this fit runs on (eager, captured or replayed).  A loop of steps wrapped in
        `with torch.cuda.stream(fit.stream):` runs without the per-step fences against the caller's stream."""
        return self._runner.stream

    @property
    def graph_captured(self):
        return self._runner.graph is not None

    def step(self):
        """forward + loss + backward + all-reduce.  Returns (loss, grad_vertices, grad_textures) of the WHOLE objective
        (all ranks' cameras) as views of the persistent flat buffer, valid until the next step."""
        nv = self.vertices.numel()
        if self.split_exchange:
            flat = self._runner()                   # ... during which the texture part's all-reduce was started
            rest = allreduce_sum_start(flat[:1 + nv])
            allreduce_wait(self._tex_work)
            allreduce_wait(rest)
            self._tex_work = None
        else:
            flat = allreduce_sum_(self._runner())
        gv = flat[1:1 + nv].view_as(self.vertices)
        gt = flat[1 + nv:].view_as(self.textures) if self.textures.requires_grad else None
        return flat[0], gv, gt
