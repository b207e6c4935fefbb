"""Domain-specific BatchNorm -- MI355X build.

The reference's networks/dsbn.py:4-33 keeps one BatchNorm2d per domain in a ModuleList called `bns` and, on a call, uses
`self.bns[domain_label[0]]` for the whole batch (dsbn.py:24-27), returning `(output, domain_label)`.  It is reached only through
networks/unet.py, which no script imports.  Here the class is the OPTIONAL per-domain statistics of
BASELINE.json configs[2]: `UNet(..., num_domains=D)` puts one in place of every BatchNorm2d and `UNet.forward(x, domain_label=...)`
hands the selected member's parameters and running buffers to libustrun.so, where the BatchNorm arithmetic lives (ustrun.engine).
What is kept from the reference is the contract a checkpoint sees: the class name, the constructor's arguments and the member name
`bns`, hence the state_dict keys `bns.<d>.{weight,bias,running_mean,running_var,num_batches_tracked}`.
"""
import torch.nn as nn


class DomainSpecificBatchNorm2d(nn.Module):
    def __init__(self, num_features, num_domains, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        if num_domains < 1:
            raise ValueError(f"num_domains = {num_domains}")
        self.num_features, self.num_domains = num_features, num_domains
        members = []
        for _ in range(num_domains):
            members.append(nn.BatchNorm2d(num_features, eps=eps, momentum=momentum, affine=affine,
                                          track_running_stats=track_running_stats))
        self.bns = nn.ModuleList(members)

    def select(self, domain_label):
        """The member of the batch's domain: the FIRST label decides for every sample (dsbn.py:26)."""
        first = domain_label[0] if hasattr(domain_label, "__getitem__") else domain_label
        d = int(first)
        if d < 0 or d >= self.num_domains:
            raise IndexError(f"domain {d} of {self.num_domains}")
        return self.bns[d]

    def forward(self, x, domain_label):
        if x.dim() != 4:
            raise ValueError(f"expected 4D input (got {x.dim()}D input)")
        raise RuntimeError("DomainSpecificBatchNorm2d runs inside UNet.forward(x, domain_label=...) -- one fused call into libustrun.so "
                           f"with the parameters of {type(self.select(domain_label)).__name__} member {int(domain_label[0])}; a "
                           "stand-alone BatchNorm has no HIP path of its own and there is no CPU fallback")
