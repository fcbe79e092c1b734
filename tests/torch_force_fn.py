"""TEST REFERENCE ONLY: the jellyfish design gradient through the stock-torch surrogate modules + torch autograd
(/root/reference/inference/inference_2d_jellyfish.py:85-114 `force_fn`, :49-61 `reg_theta`).  The product path is
diffphycon_amd/model/surrogates_hip.py (explicit reverse pass on libdpc); nothing under diffphycon_amd/ or inference/ imports this."""
import torch

from diffphycon_amd.diffusion.diffusion_2d_jellyfish import reg_theta


def force_fn(x, bd_0, force_model, bd_updater, args):
    """inference_2d_jellyfish.py:85-114 (args: only_vis_pressure, device, reg_ratio, p_min, p_max)."""
    if args.only_vis_pressure:
        state, theta_expand = x[:, :, :1], x[:, :, -1]
    else:
        state, theta_expand = x[:, :, :3], x[:, :, 3]
    state.requires_grad_()
    theta_expand.requires_grad_()
    theta = torch.mean(torch.mean(theta_expand, dim=3), dim=2)
    pressure = state[:, :, 0] if args.only_vis_pressure else state[:, :, 2]
    pressure = (0.5 * pressure + 0.5) * (args.p_max - args.p_min) + args.p_min          # unnormalize_state :40-41
    pred_bd = bd_updater(bd_0.reshape(-1, *bd_0.shape[2:]), theta.reshape(-1)).reshape(bd_0.shape)
    inp = torch.cat((pressure.unsqueeze(2), pred_bd), dim=2)
    force = force_model(inp.reshape(-1, *inp.shape[2:])).reshape(state.shape[0], state.shape[1])
    weight = torch.arange(force.shape[1], 0, -1, dtype=torch.float32, device=force.device).expand(force.shape[0], force.shape[1])
    guidance = -torch.mean(force * weight, dim=1) + args.reg_ratio * reg_theta(theta)
    return torch.autograd.grad(guidance, [state, theta_expand], grad_outputs=torch.ones_like(guidance))


def design_fn_torch(force_model, bd_updater, args):
    """The callable the reference hands to `sample(design_fn=...)` (:652-659): cat([grad_state, grad_theta[:, :, None]], dim=2)."""
    def design_fn(x, bd_0):
        with torch.enable_grad():
            gs, gt = force_fn(x.clone().detach().requires_grad_(), bd_0, force_model, bd_updater, args)
        return torch.cat([gs, gt.unsqueeze(2)], dim=2)
    design_fn.analytic = True        # accepted by GaussianDiffusion._design: it returns the finished gradient, like HipDesignGradient
    return design_fn
