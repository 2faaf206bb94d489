"""The reference's model classes over the HIP layers: ``Model``, ``MLP`` and
``GCN`` with the reference's constructor keywords, attributes and method names
(gcn/models.py:8-177); ``gcn_cheby`` is ``GCN`` fed 1 + K supports
(gcn/train_normal.py:37-40).

A model is a stack described by data: each concrete class names its layer type
and the stack is laid out from ``(input_dim, FLAGS.hidden1, output_dim)``.
Variables are collected from the layers that own them, the weight-decay terms of
the loss are handed to the optimizer as decay coefficients (their gradient is
added inside the Adam kernel), and checkpoints are plain tensor files.
"""
from . import tfcompat as tf
from .layers import *  # noqa: F401,F403  (the reference's models module re-exports the layer names)
from .layers import Dense, GraphConvolution
from .metrics import *  # noqa: F401,F403
from .metrics import masked_accuracy, masked_softmax_cross_entropy

flags = tf.app.flags
FLAGS = flags.FLAGS

_MODEL_KEYWORDS = ('name', 'logging')


def _identity(x):
    return x


class Model(object):
    """Base class (reference gcn/models.py:8-82).  Subclasses fill ``layers`` in
    ``_build`` and define ``_loss`` / ``_accuracy``; ``build`` wires the stack.

    Attributes kept from the reference: ``name``, ``logging``, ``vars``
    ({variable name: variable}), ``placeholders``, ``layers``, ``activations``
    (input followed by every layer's output), ``inputs``, ``outputs``, ``loss``,
    ``accuracy``, ``optimizer``, ``opt_op``.
    """

    def __init__(self, **kwargs):
        for key in kwargs:
            assert key in _MODEL_KEYWORDS, 'Invalid keyword argument: ' + key
        self.name = kwargs.get('name') or type(self).__name__.lower()
        self.logging = kwargs.get('logging', False)
        self.vars, self.placeholders = {}, {}
        self.layers, self.activations = [], []
        self.inputs = self.outputs = None
        self.loss = self.accuracy = 0
        self.optimizer = self.opt_op = None

    # ---- hooks of the subclasses
    def _build(self):
        raise NotImplementedError

    def _loss(self):
        raise NotImplementedError

    def _accuracy(self):
        raise NotImplementedError

    def predict(self):
        pass

    # ---- wiring
    def build(self):
        """Layers (created under the model's variable scope), the chain of their
        outputs, then loss, accuracy and the training op
        (reference gcn/models.py:38-58)."""
        created_before = len(tf.get_collection(tf.GraphKeys.GLOBAL_VARIABLES))
        with tf.variable_scope(self.name):
            self._build()
        signal = self.inputs
        self.activations = [signal]
        for layer in self.layers:
            signal = layer(signal)
            self.activations.append(signal)
        self.outputs = signal
        # The variables of a model: the reference collects the global variables under the
        # model's scope (gcn/models.py:50-51).  Here: those its layers created, plus any
        # variable a subclass's _build created under the model's scope outside a layer
        # (only variables born during this build, so that a second model of the same
        # name does not pick up the first one's) -- the set that save / load see is the
        # set the optimizer trains.
        self.vars = {v.name: v for layer in self.layers for v in layer.vars.values()}
        in_scope = tf.get_collection(tf.GraphKeys.GLOBAL_VARIABLES, scope=self.name + "/")
        born_here = set(id(v) for v in tf.get_collection(tf.GraphKeys.GLOBAL_VARIABLES)[created_before:])
        for v in in_scope:
            if id(v) in born_here:
                self.vars.setdefault(v.name, v)
        self._loss()
        self._accuracy()
        self.opt_op = self.optimizer.minimize(self.loss)

    # ---- checkpoints (reference gcn/models.py:69-82: tmp/<name>.ckpt)
    def _checkpoint(self, sess):
        if not sess:
            raise AttributeError("TensorFlow session not provided.")
        return tf.train.Saver(self.vars), "tmp/%s.ckpt" % self.name

    def save(self, sess=None):
        saver, path = self._checkpoint(sess)
        print("Model saved in file: %s" % saver.save(sess, path))

    def load(self, sess=None):
        saver, path = self._checkpoint(sess)
        saver.restore(sess, path)
        print("Model restored from file: %s" % path)


class _TwoLayerClassifier(Model):
    """What ``MLP`` and ``GCN`` share (reference gcn/models.py:85-177): two
    layers of one type, relu on sparse inputs then identity, dropout in front of
    both; loss = weight decay on the FIRST layer's variables only + masked
    softmax cross-entropy; Adam at ``FLAGS.learning_rate``."""

    layer_type = None

    def __init__(self, placeholders, input_dim, **kwargs):
        super(_TwoLayerClassifier, self).__init__(**kwargs)
        self.placeholders = placeholders
        self.inputs = placeholders['features']
        self.input_dim = input_dim
        self.output_dim = placeholders['labels'].get_shape().as_list()[1]
        self.optimizer = tf.train.AdamOptimizer(learning_rate=FLAGS.learning_rate)
        self.build()

    def _build(self):
        widths = (self.input_dim, FLAGS.hidden1, self.output_dim)
        for depth in range(2):
            first = depth == 0
            self.layers.append(self.layer_type(
                input_dim=widths[depth], output_dim=widths[depth + 1], placeholders=self.placeholders,
                act=tf.nn.relu if first else _identity, dropout=True, sparse_inputs=first,
                logging=self.logging))

    def _loss(self):
        decay = sum(FLAGS.weight_decay * tf.nn.l2_loss(w) for w in self.layers[0].vars.values())
        self.loss = decay + masked_softmax_cross_entropy(
            self.outputs, self.placeholders['labels'], self.placeholders['labels_mask'])

    def _accuracy(self):
        self.accuracy = masked_accuracy(
            self.outputs, self.placeholders['labels'], self.placeholders['labels_mask'])

    def predict(self):
        return tf.nn.softmax(self.outputs)


class MLP(_TwoLayerClassifier):
    """Two ``Dense`` layers (reference gcn/models.py:85-129)."""
    layer_type = Dense


class GCN(_TwoLayerClassifier):
    """Two ``GraphConvolution`` layers (reference gcn/models.py:132-177)."""
    layer_type = GraphConvolution
