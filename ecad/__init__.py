"""Import alias: ``ecad.<module>`` resolves to ``ecad_amd.<module>``.

The reference's entry points and user code import ``ecad.image_generators...``, ``ecad.schedulers...`` etc.; with this
alias on the path instead of the reference's own ``ecad`` package those imports land on the MI355X-native host side
unchanged (same class names, constructor keywords and JSON formats -- see INTEGRATION.md).
"""
import importlib
import importlib.abc
import importlib.util
import sys


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith("ecad."):
            return None
        real = "ecad_amd." + fullname[len("ecad."):]
        try:
            if importlib.util.find_spec(real) is None:
                return None
        except ModuleNotFoundError:
            return None
        return importlib.util.spec_from_loader(fullname, self)

    def create_module(self, spec):
        mod = importlib.import_module("ecad_amd." + spec.name[len("ecad."):])
        sys.modules[spec.name] = mod
        return mod

    def exec_module(self, module):
        pass


sys.meta_path.insert(0, _AliasFinder())
