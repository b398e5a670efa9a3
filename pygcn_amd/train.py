"""Semi-supervised node classification with the 2-layer GCN on the MI355X path; run with
cwd = this directory, like the reference (`python train.py`).

Command-line surface = the reference's (pygcn/train.py:36-51: --no-cuda --fastmode --seed --epochs
--lr --weight_decay --hidden --dropout; `--hidden` keeps upstream's default 16, which the fork
retains in a comment at train.py:48), plus --path/--dataset for the citation files.  One epoch has
upstream semantics, which the fork preserves as comments (train.py:140-157): Adam step on the NLL
of the idx_train rows, validation on idx_val, final test on idx_test; the per-epoch log line has
the same fields.  The fork's own epoch body (SafeGraph samples, gradient accumulation, MLP head)
is out of scope.
"""
import argparse
import time

import numpy as np
import torch

from functional import nll_loss      # F.nll_loss(mean) of train.py:153 as a gather / scatter pair

from models import GCN
from utils import DEFAULT_CORA, accuracy, load_data

# (flag, type, default, help) — value flags; the two switches follow
VALUE_FLAGS = (
    ("--seed", int, 42, "Random seed."),
    ("--epochs", int, 200, "Number of epochs to train."),
    ("--lr", float, 0.01, "Initial learning rate."),
    ("--weight_decay", float, 5e-4, "Weight decay (L2 loss on parameters)."),
    ("--hidden", int, 16, "Number of hidden units."),
    ("--dropout", float, 0.5, "Dropout rate (1 - keep probability)."),
    ("--path", str, DEFAULT_CORA, "Directory with <dataset>.cites[/.content], a .cites file, or "
                                  "the committed edge-list fixture."),
    ("--dataset", str, "cora", "Dataset name inside --path."),
)
SWITCHES = (
    ("--no-cuda", "Disables CUDA training (unsupported here: the path is HIP-only)."),
    ("--fastmode", "Validate during training pass."),
)


def build_parser():
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    for flag, help_text in SWITCHES:
        ap.add_argument(flag, action="store_true", default=False, help=help_text)
    for flag, kind, default, help_text in VALUE_FLAGS:
        ap.add_argument(flag, type=kind, default=default, help=help_text)
    return ap


parser = build_parser()


class Run:
    """Model, optimizer and the device-resident data of one training run."""

    def __init__(self, args):
        for seed_fn in (np.random.seed, torch.manual_seed, torch.cuda.manual_seed):
            seed_fn(args.seed)
        data = load_data(args.path, args.dataset)
        self.adj, self.features, self.labels = (t.cuda() for t in data[:3])
        self.split = dict(zip(("train", "val", "test"), (t.cuda() for t in data[3:])))
        self.model = GCN(nfeat=self.features.shape[1], nhid=args.hidden,
                         nclass=int(self.labels.max().item()) + 1, dropout=args.dropout).cuda()
        self.opt = torch.optim.Adam(self.model.parameters(), lr=args.lr,
                                    weight_decay=args.weight_decay)
        self.fastmode = args.fastmode

    def score(self, log_probs, name):
        rows = self.split[name]
        return nll_loss(log_probs[rows], self.labels[rows]), accuracy(log_probs[rows], self.labels[rows])

    def epoch(self, number):
        started = time.time()
        self.model.train()
        self.opt.zero_grad()
        # upstream: output = model(features, adj); loss_train = F.nll_loss(output[idx_train], ...)
        # (train.py:150-153).  Same forward pass; the model is told which rows the loss reads so
        # that the backward pass stays on the rows that can be non-zero (pygcn_amd/fused.py)
        rows = self.split["train"]
        train_rows, log_probs = self.model(self.features, self.adj, rows=rows, keep_full=True)
        loss, acc = nll_loss(train_rows, self.labels[rows]), accuracy(train_rows, self.labels[rows])
        loss.backward()
        self.opt.step()
        if not self.fastmode:          # upstream re-evaluates with dropout off for validation
            self.model.eval()
            log_probs = self.model(self.features, self.adj)
        val_loss, val_acc = self.score(log_probs, "val")
        print("Epoch: %04d loss_train: %.4f acc_train: %.4f loss_val: %.4f acc_val: %.4f time: %.4fs"
              % (number, loss.item(), acc.item(), val_loss.item(), val_acc.item(),
                 time.time() - started))

    def test(self):
        self.model.eval()
        loss, acc = self.score(self.model(self.features, self.adj), "test")
        print("Test set results: loss= %.4f accuracy= %.4f" % (loss.item(), acc.item()))


def main():
    args = parser.parse_args()
    if args.no_cuda or not torch.cuda.is_available():
        raise SystemExit("pygcn_amd runs on an MI355X (HIP) device only; there is no CPU path.")
    run = Run(args)
    started = time.time()
    for number in range(1, args.epochs + 1):
        run.epoch(number)
    print("Optimization Finished!")
    print("Total time elapsed: %.4fs" % (time.time() - started))
    run.test()


if __name__ == "__main__":
    main()
