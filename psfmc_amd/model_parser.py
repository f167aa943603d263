"""
Model-file DSL (host glue, kept per BASELINE.json).

A model file is ordinary Python in which every bare expression statement that
evaluates to a component adds that component to the model, e.g.

    Configuration(obs_file='sci.fits', ...)
    Sky(adu=Normal(loc=0, scale=0.01))

Behaviour follows psfMC/model_parser.py:26-66: component and prior classes are
available without imports, the file runs with its own directory as cwd so
relative FITS names resolve, and only ComponentBase instances are collected.
Model files written for the reference import from `psfMC.ModelComponents` /
`psfMC.distributions`; those imports are redirected to this package.
"""
import ast
import os

from . import ModelComponents, distributions
from .ModelComponents.ComponentBase import ComponentBase

_COLLECT = '__components'
_REDIRECT = {'psfMC': 'psfmc_amd'}


class _Rewrite(ast.NodeTransformer):
    """bare expression  ->  __components.append(<expr>) ; psfMC.* imports ->
    psfmc_amd.*"""

    def visit_Expr(self, node):
        call = ast.Call(
            func=ast.Attribute(value=ast.Name(id=_COLLECT, ctx=ast.Load()),
                               attr='append', ctx=ast.Load()),
            args=[node.value], keywords=[])
        return ast.copy_location(ast.Expr(value=call), node)

    def visit_ImportFrom(self, node):
        if node.module and node.level == 0:
            head, _, tail = node.module.partition('.')
            if head in _REDIRECT:
                node.module = _REDIRECT[head] + ('.' + tail if tail else '')
        return node

    def visit_Import(self, node):
        for alias in node.names:
            head, _, tail = alias.name.partition('.')
            if head in _REDIRECT:
                if alias.asname is None:
                    alias.asname = head if not tail else None
                alias.name = _REDIRECT[head] + ('.' + tail if tail else '')
        return node

    # do not descend into function / class bodies: only module-level
    # expressions define components
    def visit_FunctionDef(self, node):
        return node

    def visit_ClassDef(self, node):
        return node


def component_list_from_file(filename):
    """Run a model file and return its components in file order."""
    with open(filename) as f:
        tree = ast.parse(f.read(), filename=filename)
    tree = _Rewrite().visit(tree)
    ast.fix_missing_locations(tree)
    code = compile(tree, filename, mode='exec')

    namespace = {_COLLECT: []}
    for module in (ModelComponents, distributions):
        for name in module.__all__:
            namespace[name] = getattr(module, name)

    here = os.getcwd()
    model_dir = os.path.dirname(os.path.abspath(filename))
    try:
        os.chdir(model_dir)
        exec(code, namespace)
    finally:
        os.chdir(here)
    return [c for c in namespace[_COLLECT] if isinstance(c, ComponentBase)]
