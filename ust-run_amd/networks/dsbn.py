"""Domain-specific BatchNorm -- MI355X build.

Same class names, constructor and members as the reference's networks/dsbn.py:4-33 (`bns`: one nn.BatchNorm2d per domain; a call
selects `self.bns[domain_label[0]]`, dsbn.py:24-27, and returns `(output, domain_label)`), so its state_dict keys
(`bns.<d>.weight`, `.bias`, `.running_mean`, `.running_var`, `.num_batches_tracked`) interchange.  The reference reaches this class
only through networks/unet.py (imported by no script): here it is the OPTIONAL per-domain statistics of
BASELINE.json configs[2] -- `UNet(..., num_domains=D)` puts one of these in place of every BatchNorm2d of the network and
`UNet.forward(x, domain_label=...)` hands the selected domain's parameters and running buffers to libustrun.so.  The arithmetic of
the selected BatchNorm is the fused network's (ustrun.engine): a stand-alone call of this module is not on any path and raises.
"""
from torch import nn


class _DomainSpecificBatchNorm(nn.Module):
    _version = 2

    def __init__(self, num_features, num_domains, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super(_DomainSpecificBatchNorm, self).__init__()
        self.num_features, self.num_domains = num_features, num_domains
        self.bns = nn.ModuleList(
            [nn.BatchNorm2d(num_features, eps, momentum, affine, track_running_stats) for _ in range(num_domains)])

    def reset_running_stats(self):
        for bn in self.bns:
            bn.reset_running_stats()

    def reset_parameters(self):
        for bn in self.bns:
            bn.reset_parameters()

    def _check_input_dim(self, input):
        raise NotImplementedError

    def select(self, domain_label):
        """the BatchNorm2d of the batch's domain: the FIRST label decides for the whole batch (dsbn.py:26)"""
        d = int(domain_label[0]) if hasattr(domain_label, "__getitem__") else int(domain_label)
        if not 0 <= d < len(self.bns):
            raise IndexError(f"domain {d} of {len(self.bns)}")
        return self.bns[d]

    def forward(self, x, domain_label):
        self._check_input_dim(x)
        raise RuntimeError("DomainSpecificBatchNorm2d runs inside UNet.forward(x, domain_label=...) (one fused call into libustrun.so); "
                           "a stand-alone BatchNorm has no HIP path of its own and there is no CPU fallback")


class DomainSpecificBatchNorm2d(_DomainSpecificBatchNorm):
    def _check_input_dim(self, input):
        if input.dim() != 4:
            raise ValueError('expected 4D input (got {}D input)'.format(input.dim()))
