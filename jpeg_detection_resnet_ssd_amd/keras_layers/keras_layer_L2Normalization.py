"""L2Normalization layer of the SSD head -- same constructor, weight name (`<name>_gamma`) and
semantics as localisation_part/keras_layers/keras_layer_L2Normalization.py:25-70:
output = K.l2_normalize(x, axis=3) * gamma, gamma trainable, initialised to `gamma_init` (20)."""
import numpy as np

from ..engine import Value, call, query, rows_of
from ..keras import backend as K
from ..keras.layers import InputSpec, Layer, _materialised


class L2Normalization(Layer):
    def __init__(self, gamma_init=20, **kwargs):
        if K.image_dim_ordering() == "tf":
            self.axis = 3
        else:
            self.axis = 1
        self.gamma_init = gamma_init
        super(L2Normalization, self).__init__(**kwargs)

    def build(self, input_shape):
        self.input_spec = [InputSpec(shape=input_shape)]
        from ..keras import initializers
        self.gamma = self.add_weight("{}_gamma".format(self.name), (input_shape[self.axis],),
                                     initializers.constant(self.gamma_init))
        super(L2Normalization, self).build(input_shape)

    def get_config(self):
        config = {"gamma_init": self.gamma_init}
        base_config = super(L2Normalization, self).get_config()
        return dict(list(base_config.items()) + list(config.items()))

    def runs_beside(self, plan, model, ins):
        """Forward launch on the side stream (engine.Plan.emit_side) when every reader is a predictor head that runs there."""
        x = ins[0]
        if not plan.forward_side_ok(x) or x.is_affine or getattr(x, "pending_add", None) is not None:
            return False
        users = model.consumers_of(self.outbound[0])
        from ..keras.layers import Conv2D
        return bool(users) and all(type(u) is Conv2D and u.activation is None for u in users)

    def lower(self, plan, model, ins):
        x = ins[0]
        xbuf = _materialised(x, self.name, plan)
        rows, c, ldx = rows_of(xbuf)
        y = plan.empty(*xbuf.shape)
        rnorm = plan.empty(rows)
        gamma = self.gamma.param
        out = Value(y, needs_grad=True, name=self.name)
        if self.runs_beside(plan, model, ins) and xbuf is x.buf:
            # only predictor heads read this: normalise beside the main chain, in front of them on the side stream
            plan.emit_side(lambda: call("dj_l2norm_fwd", xbuf, ldx, gamma, y, c, rnorm, rows, c), x.ready_event)
            out.ready_event = x.ready_event     # the heads queue behind this launch on the same stream
            plan.side_results([out])
        else:
            plan.emit(lambda: call("dj_l2norm_fwd", xbuf, ldx, gamma, y, c, rnorm, rows, c))

        def build_backward():
            if out.grad is None:
                return
            assert out.grad.mask_y is None
            dy = out.grad.buf
            ld_dy = rows_of(dy)[2]
            dx, beta, ld_dx = None, 0, 0
            if x.needs_grad:
                dx, beta = plan.grad_of(x)
                ld_dx = rows_of(dx)[2]
            partial, nr = None, 0
            if self.gamma.trainable:
                nr = query("dj_reduce_rows", rows)
                partial = plan.empty(nr, 2, c)
            plan.emit_bwd(lambda: call("dj_l2norm_bwd", dy, ld_dy, xbuf, ldx, gamma, rnorm, dx, ld_dx, partial, rows, c,
                                       beta))
            if partial is not None:
                dg = self.gamma.grad
                plan.emit_bwd(lambda: call("dj_colreduce_finalize", partial, nr, c, 0, dg, 0))
                plan.note_grad(self.gamma)

        plan.on_backward(build_backward)
        return out
