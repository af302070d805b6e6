"""Command-line driver with the reference's flags (reference main.py:9-14):

    python main.py [--pretrain] [--train] [--restart] --config_path=experiments/<name>.cfg

Data parallel over the GPUs of one node: launch one process per GPU, e.g.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        main.py --train --config_path=experiments/no_unfreezing.cfg
"""
import argparse

import numpy as np
import torch

from data import get_ASR_datasets, get_SLU_datasets, read_config
from models import PretrainedModel, Model, set_dropout_seed
from slu_hip import dp
from training import Trainer


def run(args):
    rank, world_size, local_rank = dp.init_from_env()
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    config = read_config(args.config_path)
    torch.manual_seed(config.seed)
    np.random.seed(config.seed)
    set_dropout_seed(config.seed + 7919 * rank)          # independent dropout streams per rank
    say = print if rank == 0 else (lambda *a, **k: None)

    if args.pretrain:
        train_dataset, valid_dataset, test_dataset = get_ASR_datasets(config)
        pretrained_model = PretrainedModel(config=config)
        trainer = Trainer(model=pretrained_model, config=config)
        if args.restart:
            trainer.load_checkpoint()
        n = config.pretraining_num_epochs
        for epoch in range(n):
            say("========= Epoch %d of %d =========" % (epoch + 1, n))
            tr = trainer.train(train_dataset)
            va = trainer.test(valid_dataset)
            say("========= Results: epoch %d of %d =========" % (epoch + 1, n))
            say("*phonemes*| train accuracy: %.2f| train loss: %.2f| valid accuracy: %.2f| valid loss: %.2f\n"
                % (tr[0], tr[1], va[0], va[1]))
            say("*words*| train accuracy: %.2f| train loss: %.2f| valid accuracy: %.2f| valid loss: %.2f\n"
                % (tr[2], tr[3], va[2], va[3]))
            trainer.save_checkpoint()
        trainer.close()

    if args.train:
        train_dataset, valid_dataset, test_dataset = get_SLU_datasets(config)
        model = Model(config=config)
        trainer = Trainer(model=model, config=config)
        if args.restart:
            trainer.load_checkpoint()
        n = config.training_num_epochs
        valid_acc = valid_loss = float("nan")
        for epoch in range(n):
            say("========= Epoch %d of %d =========" % (epoch + 1, n))
            train_acc, train_loss = trainer.train(train_dataset)
            valid_acc, valid_loss = trainer.test(valid_dataset)
            say("========= Results: epoch %d of %d =========" % (epoch + 1, n))
            say("*intents*| train accuracy: %.2f| train loss: %.2f| valid accuracy: %.2f| valid loss: %.2f\n"
                % (train_acc, train_loss, valid_acc, valid_loss))
            trainer.save_checkpoint()
        test_acc, test_loss = trainer.test(test_dataset)
        say("========= Test results =========")
        say("*intents*| test accuracy: %.2f| test loss: %.2f| valid accuracy: %.2f| valid loss: %.2f\n"
            % (test_acc, test_loss, valid_acc, valid_loss))
        trainer.close()


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--pretrain", action="store_true", help="run ASR pre-training")
    parser.add_argument("--train", action="store_true", help="run SLU training")
    parser.add_argument("--restart", action="store_true", help="load checkpoint from a previous run")
    parser.add_argument("--config_path", type=str, help="path to config file with hyperparameters, etc.")
    run(parser.parse_args())
