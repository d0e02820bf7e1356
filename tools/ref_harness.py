"""Import the UNMODIFIED reference (/root/reference) in this container, for fixture generation only.

The reference needs hydra / omegaconf / pytorch_lightning / calvin_agent, none of which are installed;
this module installs minimal ``sys.modules`` stand-ins *for the import machinery only* (attribute dicts,
``instantiate`` by dotted path, an ``nn.Module``-based LightningModule).  None of the reference's
arithmetic is touched or restated here.  Never shipped to the GPU box's test path: the fixtures under
``tests/golden`` are what travels.
"""
from __future__ import annotations

import importlib
import sys
import types

sys.dont_write_bytecode = True
REF = "/root/reference"


class DictConfig(dict):
    """Attribute dict standing in for omegaconf.DictConfig."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:  # pragma: no cover
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


class ListConfig(list):
    pass


def to_cfg(x):
    if isinstance(x, dict):
        return DictConfig({k: to_cfg(v) for k, v in x.items()})
    return x


def instantiate(cfg, *args, **kwargs):
    if cfg is None or (isinstance(cfg, dict) and len(cfg) == 0):
        return None
    cfg = dict(cfg)
    target = cfg.pop("_target_")
    cfg.pop("_recursive_", None)
    mod, _, name = target.rpartition(".")
    fn = getattr(importlib.import_module(mod), name)
    cfg.update(kwargs)
    return fn(*args, **cfg)


def install_stubs():
    import torch.nn as nn

    if "omegaconf" in sys.modules and getattr(sys.modules["omegaconf"], "_hulc_stub", False):
        return
    om = types.ModuleType("omegaconf")
    om._hulc_stub = True
    om.DictConfig = DictConfig
    om.ListConfig = ListConfig

    class OmegaConf:
        @staticmethod
        def load(path):
            raise FileNotFoundError(path)

    om.OmegaConf = OmegaConf
    sys.modules["omegaconf"] = om

    hy = types.ModuleType("hydra")
    hyu = types.ModuleType("hydra.utils")
    hyu.instantiate = instantiate
    hy.utils = hyu
    sys.modules["hydra"] = hy
    sys.modules["hydra.utils"] = hyu

    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(nn.Module):
        def __init__(self):
            super().__init__()
            self.logged = {}

        @property
        def device(self):
            import torch

            return torch.device("cpu")

        def log(self, name, value, **kw):
            self.logged[name] = float(value)

        def save_hyperparameters(self, *a, **k):
            pass

    pl.LightningModule = LightningModule
    pl.Trainer = object
    pl.Callback = object
    plu = types.ModuleType("pytorch_lightning.utilities")
    plu.rank_zero_only = lambda f: f
    plu.rank_zero_info = lambda *a, **k: None
    pl.utilities = plu
    sys.modules["pytorch_lightning"] = pl
    sys.modules["pytorch_lightning.utilities"] = plu

    ca = types.ModuleType("calvin_agent")
    cam = types.ModuleType("calvin_agent.models")
    cab = types.ModuleType("calvin_agent.models.calvin_base_model")

    class CalvinBaseModel:
        pass

    cab.CalvinBaseModel = CalvinBaseModel
    sys.modules["calvin_agent"] = ca
    sys.modules["calvin_agent.models"] = cam
    sys.modules["calvin_agent.models.calvin_base_model"] = cab
    if REF not in sys.path:
        sys.path.insert(0, REF)


def model_cfg(kind="hulc", max_window=32, use_clip=True, dropout_p=0.1):
    """The resolved conf/model/{hulc,gcbc}.yaml tree (values copied from conf/**)."""
    vf = 64
    cfg = dict(
        perceptual_encoder=dict(
            _target_="hulc.models.perceptual_encoders.concat_encoders.ConcatEncoders",
            rgb_static=dict(
                _target_="hulc.models.perceptual_encoders.vision_network.VisionNetwork",
                input_width=200, input_height=200, activation_function="ReLU", dropout_vis_fc=0.0,
                l2_normalize_output=False, visual_features=vf, num_c=3, use_sinusoid=False,
                spatial_softmax_temp=1.0),
            rgb_gripper=dict(
                _target_="hulc.models.perceptual_encoders.vision_network_gripper.VisionNetwork",
                input_width=84, input_height=84, activation_function="ReLU", dropout_vis_fc=0.0,
                l2_normalize_output=False, visual_features=vf, conv_encoder="nature_cnn", num_c=3),
            depth_static={}, depth_gripper={}, proprio={}, tactile={}),
        plan_proposal=dict(
            _target_="hulc.models.plan_encoders.plan_proposal_net.PlanProposalNetwork",
            perceptual_features=None, latent_goal_features=32, plan_features=None,
            activation_function="ReLU", hidden_size=2048),
        plan_recognition=dict(
            _target_="hulc.models.plan_encoders.plan_recognition_net.PlanRecognitionTransformersNetwork",
            num_heads=8, num_layers=2, encoder_hidden_size=2048, fc_hidden_size=4096, in_features=None,
            plan_features=None, action_space=7, dropout_p=dropout_p, encoder_normalize=False,
            positional_normalize=False, position_embedding=True, max_position_embeddings=max_window),
        distribution=dict(_target_="hulc.utils.distributions.Distribution", dist="discrete",
                          category_size=32, class_size=32),
        visual_goal=dict(_target_="hulc.models.encoders.goal_encoders.VisualGoalEncoder", in_features=None,
                         hidden_size=2048, latent_goal_features=32, l2_normalize_goal_embeddings=False,
                         activation_function="ReLU"),
        language_goal=dict(_target_="hulc.models.encoders.goal_encoders.LanguageGoalEncoder", in_features=384,
                           hidden_size=2048, latent_goal_features=32, l2_normalize_goal_embeddings=False,
                           activation_function="ReLU", word_dropout_p=0.0),
        action_decoder=dict(
            _target_="hulc.models.decoders.logistic_decoder_rnn.LogisticDecoderRNN",
            n_mixtures=10, hidden_size=2048, out_features=7, log_scale_min=-7.0,
            act_max_bound=[1.0] * 7, act_min_bound=[-1.0] * 7, dataset_dir="", load_action_bounds=False,
            num_classes=10, latent_goal_features=32, plan_features=None, perceptual_features=None,
            gripper_alpha=1.0, perceptual_emb_slice=[64, 128], policy_rnn_dropout_p=0.0, num_layers=2,
            rnn_model="rnn_decoder", gripper_control=True, discrete_gripper=True),
        kl_beta=0.01, kl_balancing_mix=0.8, state_recons=False, state_recon_beta=0.5,
        use_bc_z_auxiliary_loss=False, bc_z_auxiliary_loss_beta=1.0, use_mia_auxiliary_loss=False,
        mia_auxiliary_loss_beta=1.0,
        optimizer=dict(_target_="torch.optim.Adam", lr=2e-4),
        lr_scheduler=dict(_target_="transformers.get_constant_schedule"),
        val_instructions={}, use_clip_auxiliary_loss=use_clip, clip_auxiliary_loss_beta=3.0, replan_freq=30,
        bc_z_lang_decoder=None, mia_lang_discriminator=None,
        proj_vis_lang=dict(_target_="hulc.models.auxiliary_loss_networks.proj_vis_lang.ProjVisLang",
                           im_dim=4096, lang_dim=32, output_dim=32, proj_lang=True),
    )
    return to_cfg(cfg)


def mcil_cfg(cfg, rnn_type="nn.RNN"):
    """conf/model/mcil.yaml: plan_recognition birnn, distribution continuous, action_decoder mcil_default, no CLIP loss."""
    cfg["plan_recognition"] = to_cfg(dict(_target_="hulc.models.plan_encoders.plan_recognition_net.PlanRecognitionBiRNNNetwork", in_features=None,
                                          plan_features=256, action_space=7, birnn_dropout_p=0.0, rnn_type=rnn_type))
    cfg["distribution"] = to_cfg(dict(_target_="hulc.utils.distributions.Distribution", dist="continuous", plan_features=256))
    ad = dict(cfg["action_decoder"])
    ad.update(num_classes=256, gripper_control=False, discrete_gripper=False)
    ad.pop("perceptual_emb_slice", None)
    cfg["action_decoder"] = to_cfg(ad)
    cfg["use_clip_auxiliary_loss"] = False
    cfg["proj_vis_lang"] = None
    return cfg


def build_reference(kind="hulc", **kw):
    install_stubs()
    if kind in ("mcil", "mcil_gru"):
        kw = dict(kw); kw["use_clip"] = False
        cfg = mcil_cfg(model_cfg("hulc", **kw), "nn.GRU" if kind == "mcil_gru" else "nn.RNN")
        from hulc.models.hulc import Hulc
        return Hulc(**cfg)
    cfg = model_cfg(kind, **kw)
    if kind == "hulc":
        from hulc.models.hulc import Hulc as Cls
    elif kind == "gcbc":
        from hulc.models.gcbc import GCBC as Cls
    else:
        raise ValueError(kind)
    return Cls(**cfg)


if __name__ == "__main__":
    import warnings

    warnings.filterwarnings("ignore")
    m = build_reference("hulc")
    tot = 0
    for n, p in m.named_parameters():
        print(n, tuple(p.shape))
        tot += p.numel()
    print("total", tot)
    for n, b in m.named_buffers():
        print("BUF", n, tuple(b.shape))
