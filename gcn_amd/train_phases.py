"""Per-phase trainer: every epoch runs layer-1 update, layer-1 aggregation,
layer-2 update and layer-2 aggregation as four separate ``sess.run`` calls of
the fork's split stages (``GraphConvolution._update`` / ``_aggregate``), times
each on the GPU, then does the ordinary loss / optimiser / validation runs --
the loop of reference gcn/train_forward_a_u_perf.py:132-190, with HIP-event
timing in place of ``perf stat`` child processes.  It also emits the fork's
analytic work counts (gcn/train_forward_a_u_tf_flops_profiler.py:82-118):
update FLOPs = N*F*H and N*H*C, aggregate accesses = adj.nnz*F and adj.nnz*H,
AMIR = sum(aggregate) / sum(update).

    python -m gcn_amd.train_phases --dataset cora --epochs 20 --report phases.json
"""
from __future__ import division
from __future__ import print_function

import json
import time

import numpy as np
import torch

from . import tfcompat as tf
from .train import define_flags
from .utils import construct_feed_dict, load_data, preprocess_adj, preprocess_features

FLAGS = tf.app.flags.FLAGS

PHASES = ("layer1_update", "layer1_aggregation", "layer2_update", "layer2_aggregation")


def analytic_counts(adj, features, hidden_dim, output_dim):
    """The fork's formulas, verbatim in meaning
    (gcn/train_forward_a_u_tf_flops_profiler.py:88-107,114-117)."""
    n, f = adj.shape[0], features[2][1]
    c = {
        "layer1_update": n * f * hidden_dim,
        "layer1_aggregate": adj.nnz * f,
        "layer2_update": n * hidden_dim * output_dim,
        "layer2_aggregate": adj.nnz * hidden_dim,
    }
    upd = c["layer1_update"] + c["layer2_update"]
    c["amir"] = (c["layer1_aggregate"] + c["layer2_aggregate"]) / upd if upd > 0 else 0
    return c


def run(data=None, verbose=True):
    from .models import GCN
    np.random.seed(FLAGS.seed)
    tf.reset_default_graph()
    tf.set_random_seed(FLAGS.seed)
    adj, features, y_train, y_val, y_test, train_mask, val_mask, test_mask = \
        data if data is not None else load_data(FLAGS.dataset)
    features = preprocess_features(features)
    support = [preprocess_adj(adj)]
    placeholders = {
        'support': [tf.sparse_placeholder(tf.float32)],
        'features': tf.sparse_placeholder(tf.float32, shape=tf.constant(features[2], dtype=tf.int64)),
        'labels': tf.placeholder(tf.float32, shape=(None, y_train.shape[1])),
        'labels_mask': tf.placeholder(tf.int32),
        'dropout': tf.placeholder_with_default(0., shape=()),
        'num_features_nonzero': tf.placeholder(tf.int32)
    }
    model = GCN(placeholders, input_dim=features[2][1], logging=True)
    sess = tf.Session()
    sess.run(tf.global_variables_initializer())

    # the four stage graphs are built once (the reference rebuilds them every epoch)
    feat_in = placeholders['features']
    mid1 = tf.placeholder(tf.float32)
    mid2 = tf.placeholder(tf.float32)
    mid3 = tf.placeholder(tf.float32)
    stages = [model.layers[0]._update(feat_in), model.layers[0]._aggregate(mid1),
              model.layers[1]._update(mid2), model.layers[1]._aggregate(mid3)]
    feeds_in = [None, mid1, mid2, mid3]

    def timed(node, feed):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = sess.run([node], feed_dict=feed)[0]
        e.record()
        e.synchronize()
        return out, s.elapsed_time(e)

    counts = analytic_counts(adj, features, FLAGS.hidden1, y_train.shape[1])
    per_epoch, cost_val = [], []
    for epoch in range(FLAGS.epochs):
        t = time.time()
        feed_dict = construct_feed_dict(features, support, y_train, train_mask, placeholders)
        feed_dict.update({placeholders['dropout']: FLAGS.dropout})
        rec, x = {}, None
        for name, node, ph in zip(PHASES, stages, feeds_in):
            fd = dict(feed_dict)
            if ph is not None:
                fd[ph] = x
            x, ms = timed(node, fd)
            rec[name + "_ms"] = ms
        loss_acc = sess.run([model.loss, model.accuracy], feed_dict=feed_dict)
        sess.run([model.opt_op], feed_dict=feed_dict)
        fdv = construct_feed_dict(features, support, y_val, val_mask, placeholders)
        cost, acc = sess.run([model.loss, model.accuracy], feed_dict=fdv)
        cost_val.append(cost)
        rec.update(train_loss=float(loss_acc[0]), train_acc=float(loss_acc[1]), val_loss=float(cost),
                   val_acc=float(acc), time=time.time() - t)
        per_epoch.append(rec)
        if verbose:
            print("Epoch:", '%04d' % (epoch + 1), "train_loss=", "{:.5f}".format(loss_acc[0]),
                  "train_acc=", "{:.5f}".format(loss_acc[1]), "val_loss=", "{:.5f}".format(cost),
                  "val_acc=", "{:.5f}".format(acc), "AMIR=", "{:.4f}".format(counts["amir"]),
                  " ".join("%s=%.3fms" % (p, rec[p + "_ms"]) for p in PHASES),
                  "time=", "{:.5f}".format(time.time() - t))
        if epoch > FLAGS.early_stopping and cost_val[-1] > np.mean(cost_val[-(FLAGS.early_stopping + 1):-1]):
            if verbose:
                print("Early stopping...")
            break
    summary = {
        "dataset_stats": {"num_nodes": int(adj.shape[0]), "feature_dim": int(features[2][1]),
                          "hidden_dim": int(FLAGS.hidden1), "output_dim": int(y_train.shape[1]),
                          "adj_nnz": int(adj.nnz)},
        "total_flops": {k: v for k, v in counts.items() if k != "amir"},
        "total_amir": counts["amir"],
        "phase_ms_median": {p: float(np.median([r[p + "_ms"] for r in per_epoch])) for p in PHASES},
        "epochs": per_epoch,
        "last_stage_output": x,
    }
    if FLAGS.report:
        with open(FLAGS.report, "w") as fh:
            json.dump({k: v for k, v in summary.items() if k != "last_stage_output"}, fh, indent=1)
    return summary


def main():
    define_flags()
    run()


if __name__ == '__main__':
    main()
