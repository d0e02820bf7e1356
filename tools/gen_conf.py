"""Write the repo's own Hydra-style conf/ tree (same group / option names and keys as the reference's conf/ for the hot path,
SURVEY.md §8b) from the resolved values below.  Run once: python tools/gen_conf.py"""
import os

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONF = os.path.join(ROOT, "conf")


def w(rel, body, header=None):
    p = os.path.join(CONF, rel)
    os.makedirs(os.path.dirname(p), exist_ok=True)
    with open(p, "w") as f:
        if header:
            f.write("".join(f"# {l}\n" for l in header.splitlines()))
        f.write(body if isinstance(body, str) else yaml.safe_dump(body, sort_keys=True, default_flow_style=None, width=110))


H = "hulc_amd conf tree — mirrors the reference option `{}` (keys/values = the drop-in contract; engine keys are additions)."

w("config.yaml", """defaults:
  - model: hulc
  - loss: default
  - training: default_training
  - trainer: mi355x
  - datamodule: synthetic
  - callbacks: default
  - _self_
seed: 42
# one directory per run, like the reference's hydra.run.dir (runs/<date>/<time>): a re-launched command starts fresh;
# resume=true re-enters the newest run directory that holds a checkpoint (requeued jobs), log_dir=<run dir> resumes that run
log_dir: ./runs/{now}
resume: false
""", H.format("conf/config.yaml"))

for name, tgt in (("hulc", "hulc.models.hulc.Hulc"), ("gcbc", "hulc.models.gcbc.GCBC")):
    w(f"model/{name}.yaml", f"""defaults:
  - perceptual_encoder: gripper_cam
  - plan_proposal: default
  - plan_recognition: transformers
  - distribution: discrete
  - visual_goal: default
  - language_goal: default
  - action_decoder: hulc_default
  - optimizer: adam
  - lr_scheduler: constant
  - proj_vis_lang: default
_target_: {tgt}
_recursive_: false
# loss weights are interpolated from the loss group exactly like the reference
kl_beta: ${{loss.kl_beta}}
kl_balancing_mix: ${{loss.kl_balancing_mix}}
clip_auxiliary_loss_beta: ${{loss.clip_auxiliary_loss_beta}}
state_recon_beta: ${{loss.state_recon_beta}}
bc_z_auxiliary_loss_beta: ${{loss.bc_z_auxiliary_loss_beta}}
mia_auxiliary_loss_beta: ${{loss.mia_auxiliary_loss_beta}}
use_clip_auxiliary_loss: true
state_recons: false
use_bc_z_auxiliary_loss: false
use_mia_auxiliary_loss: false
replan_freq: 30
bc_z_lang_decoder: null
mia_lang_discriminator: null
val_instructions: {{}}
# ---- engine additions (hulc_amd.Hulc keyword arguments)
precision: ${{trainer.precision}}
max_batch_size: ${{datamodule.batch_size}}
seed: ${{seed}}
""", H.format(f"conf/model/{name}.yaml"))

w("model/perceptual_encoder/gripper_cam.yaml", """defaults:
  - rgb_static: default
  - rgb_gripper: default
_target_: hulc.models.perceptual_encoders.concat_encoders.ConcatEncoders
_recursive_: false
depth_static: {}
depth_gripper: {}
proprio: {}
tactile: {}
""", H.format("conf/model/perceptual_encoder/gripper_cam.yaml"))
w("model/perceptual_encoder/rgb_static/default.yaml", dict(
    _target_="hulc.models.perceptual_encoders.vision_network.VisionNetwork", input_width=200, input_height=200, num_c=3,
    visual_features=64, activation_function="ReLU", dropout_vis_fc=0.0, l2_normalize_output=False, use_sinusoid=False,
    spatial_softmax_temp=1.0), H.format("rgb_static/default.yaml"))
w("model/perceptual_encoder/rgb_gripper/default.yaml", dict(
    _target_="hulc.models.perceptual_encoders.vision_network_gripper.VisionNetwork", input_width=84, input_height=84, num_c=3,
    visual_features=64, conv_encoder="nature_cnn", activation_function="ReLU", dropout_vis_fc=0.0, l2_normalize_output=False),
  H.format("rgb_gripper/default.yaml"))
w("model/plan_proposal/default.yaml", dict(_target_="hulc.models.plan_encoders.plan_proposal_net.PlanProposalNetwork", hidden_size=2048,
                                          activation_function="ReLU", latent_goal_features="${model.visual_goal.latent_goal_features}",
                                          perceptual_features="???", plan_features="???"), H.format("plan_proposal/default.yaml"))
w("model/plan_recognition/transformers.yaml", dict(
    _target_="hulc.models.plan_encoders.plan_recognition_net.PlanRecognitionTransformersNetwork", num_heads=8, num_layers=2,
    encoder_hidden_size=2048, fc_hidden_size=4096, dropout_p=0.1, encoder_normalize=False, positional_normalize=False,
    position_embedding=True, max_position_embeddings="${datamodule.max_window_size}", action_space="${datamodule.action_space}",
    in_features="??", plan_features="???"), H.format("plan_recognition/transformers.yaml"))
w("model/distribution/discrete.yaml", dict(_target_="hulc.utils.distributions.Distribution", dist="discrete", category_size=32, class_size=32),
  H.format("distribution/discrete.yaml"))
w("model/visual_goal/default.yaml", dict(_target_="hulc.models.encoders.goal_encoders.VisualGoalEncoder", hidden_size=2048,
                                        latent_goal_features=32, l2_normalize_goal_embeddings=False, activation_function="ReLU", in_features="???"),
  H.format("visual_goal/default.yaml"))
w("model/language_goal/default.yaml", dict(_target_="hulc.models.encoders.goal_encoders.LanguageGoalEncoder", in_features=384, hidden_size=2048,
                                          latent_goal_features=32, l2_normalize_goal_embeddings=False, activation_function="ReLU", word_dropout_p=0.0),
  H.format("language_goal/default.yaml"))
w("model/action_decoder/hulc_default.yaml", dict(
    _target_="hulc.models.decoders.logistic_decoder_rnn.LogisticDecoderRNN", n_mixtures=10, hidden_size=2048, num_layers=2,
    rnn_model="rnn_decoder", num_classes=10, log_scale_min=-7.0, gripper_alpha=1.0, policy_rnn_dropout_p=0.0, gripper_control=True,
    discrete_gripper=True, perceptual_emb_slice=[64, 128], load_action_bounds=False, out_features="${datamodule.action_space}",
    act_max_bound="${datamodule.action_max}", act_min_bound="${datamodule.action_min}", dataset_dir="${datamodule.root_data_dir}",
    latent_goal_features="${model.visual_goal.latent_goal_features}", plan_features="???", perceptual_features="???"),
  H.format("action_decoder/hulc_default.yaml"))
w("model/optimizer/adam.yaml", dict(_target_="torch.optim.Adam", lr="${training.lr}"), H.format("optimizer/adam.yaml"))
w("model/optimizer/adamw.yaml", dict(_target_="torch.optim.AdamW", lr="${training.lr}", weight_decay=1.0e-6), H.format("optimizer/adamw.yaml"))
w("model/optimizer/sgd.yaml", dict(_target_="torch.optim.SGD", lr="${training.lr}", momentum=0.9), H.format("optimizer/sgd.yaml"))
w("model/lr_scheduler/constant.yaml", dict(_target_="transformers.get_constant_schedule"), H.format("lr_scheduler/constant.yaml"))
# num_training_steps -1 = inferred from the trainer / datamodule (Hulc.num_training_steps); a float num_warmup_steps = fraction of the training steps
w("model/lr_scheduler/cosine_schedule_with_warmup.yaml", dict(_target_="transformers.get_cosine_schedule_with_warmup", num_training_steps=-1,
                                                               num_warmup_steps=0.1, num_cycles=0.5), H.format("lr_scheduler/cosine_schedule_with_warmup.yaml"))
w("model/lr_scheduler/linear_schedule_with_warmup.yaml", dict(_target_="transformers.get_linear_schedule_with_warmup", num_training_steps=-1,
                                                               num_warmup_steps=0.1), H.format("lr_scheduler/linear_schedule_with_warmup.yaml"))
w("model/proj_vis_lang/default.yaml", dict(_target_="hulc.models.auxiliary_loss_networks.proj_vis_lang.ProjVisLang",
                                          im_dim="${model.plan_recognition.fc_hidden_size}", lang_dim="${model.language_goal.latent_goal_features}",
                                          output_dim="${model.language_goal.latent_goal_features}", proj_lang=True), H.format("proj_vis_lang/default.yaml"))
w("model/proj_vis_lang/none.yaml", "{}\n", H.format("proj_vis_lang/none.yaml"))
w("loss/default.yaml", dict(kl_beta=0.01, kl_balancing_mix=0.8, state_recon_beta=0.5, bc_z_auxiliary_loss_beta=1.0, mia_auxiliary_loss_beta=1.0,
                           clip_auxiliary_loss_beta=3.0), H.format("loss/default.yaml"))
w("training/default_training.yaml", dict(lr=2.0e-4), H.format("training/default_training.yaml"))
w("trainer/mi355x.yaml", dict(devices=1, accelerator="gpu", precision="bf16", max_epochs=100, max_steps=-1, val_check_interval=1.0,
                             sync_batchnorm=False, allreduce_bucket_mb=0),
  "hulc_amd trainer group: bf16 MFMA operands (no loss scaling needed); trainer=play_trainer is the reference's group (precision 16 = "
  "fp16 + dynamic loss scaling); allreduce_bucket_mb=0 = one flat RCCL all-reduce")
w("trainer/play_trainer.yaml", dict(devices=1, accelerator="gpu", precision=16, val_check_interval=1.0, max_epochs=100, max_steps=-1, sync_batchnorm=False,
                                    allreduce_bucket_mb=0),
  H.format("conf/trainer/play_trainer.yaml") + " — precision 16 = Lightning native AMP: the fp16 engine + on-device GradScaler (hulc_scaler_*)")
w("datamodule/synthetic.yaml", dict(_target_="hulc_amd.trainer.SyntheticDataModule", root_data_dir="", action_space=7,
                                   action_max=[1.0] * 7, action_min=[-1.0] * 7, max_window_size=32, min_window_size=20, batch_size=32,
                                   modalities=["vis", "lang"], steps_per_epoch=50),
  "synthetic CALVIN-shaped windows (SURVEY.md §8d); the reference's CalvinDataModule (calvin_agent, absent) is out of scope — only its\n"
  "batch contract (hulc/models/hulc.py:395-414) and the keys the model tree interpolates are kept")
w("callbacks/default.yaml", """defaults:
  - kl_schedule: constant
  - checkpoint: all
""", H.format("callbacks/default.yaml"))
w("callbacks/kl_schedule/constant.yaml", dict(_target_="hulc_amd.trainer.KLConstantSchedule"), H.format("kl_schedule/constant.yaml"))
w("callbacks/kl_schedule/linear.yaml", dict(_target_="hulc_amd.trainer.KLLinearSchedule", start_epoch=10, end_epoch=50, max_kl_beta="${loss.kl_beta}"),
  H.format("kl_schedule/linear.yaml"))
w("callbacks/kl_schedule/sigmoid.yaml", dict(_target_="hulc_amd.trainer.KLSigmoidSchedule", start_epoch=10, end_epoch=50, max_kl_beta="${loss.kl_beta}"),
  H.format("kl_schedule/sigmoid.yaml"))
w("callbacks/checkpoint/all.yaml", dict(_target_="hulc_amd.trainer.ModelCheckpoint", save_top_k=-1, verbose=True, dirpath="saved_models",
                                       filename="{epoch}"), H.format("checkpoint/all.yaml"))
print("wrote", CONF)
